#!/bin/bash
# Same-box A/B of two builds of libfpsg_hip.so (box-to-box differences are as large as most kernel changes):
#   tools/ab_lib.sh <baseline .so> <tag> <command ...>   ->  gpurun_out/ab_<tag>_{base,new}.txt
# The baseline is loaded through FPSG_HIP_LIB (fpsg_amd/_hip.py); the second run uses the in-tree build.
base=$1; tag=$2; shift 2
mkdir -p gpurun_out
FPSG_HIP_LIB=$(realpath "$base") "$@" > gpurun_out/ab_${tag}_base.txt 2> gpurun_out/ab_${tag}_base.err || exit 1
"$@" > gpurun_out/ab_${tag}_new.txt 2> gpurun_out/ab_${tag}_new.err || exit 1
paste -d'\n' gpurun_out/ab_${tag}_base.txt gpurun_out/ab_${tag}_new.txt | awk 'NR%2==1{print "base " $0} NR%2==0{print "new  " $0}'
