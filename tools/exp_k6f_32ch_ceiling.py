"""Ceiling experiment for "K6f with 32 output channels per workgroup" (VERDICT r3 item 4), without building the kernel.

What the 32-channel form would change in K6f's channel step: ONE transformed patch feeds 72 MFMAs (two 16-channel
A-fragment sets) instead of 36; the transformed-filter slice is no longer resident (288 KB) but streamed, 18 KB per step,
by the workgroup's threads into an LDS ring, with one workgroup barrier per step; 18 A-fragment reads per step instead
of 9.  This script patches a COPY of fpsg_amd/csrc/winograd_fused.hip so that its step has exactly that instruction
stream -- every block of six MFMAs issued twice (the second time with a second set of A fragments read from LDS,
accumulating into the same registers), five 16-byte global loads + LDS stores per thread and step, one __syncthreads()
per step, the same trip count for the four waves of a workgroup -- and builds build_exp/libfpsg_hip_k6f72.so.  The
outputs of that kernel are WRONG by construction (every product is added twice, from the wrong filter slice for the second
set); only its time means anything:  time(72-MFMA step) / 2 against time(36-MFMA step) is the best case of the
32-channel form's channel loop (its output transform, 288 accumulator registers and the real slice addressing only
cost more).

    python tools/exp_k6f_32ch_ceiling.py            # here: patch + build
    FPSG_HIP_LIB=build_exp/libfpsg_hip_k6f72.so python tools/bench_k6f.py    # GPU box, beside the in-tree library
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fpsg_amd", "csrc")
OUT = os.path.join(ROOT, "build_exp")


def patch(s: str) -> str:
    # 1. a second set of A fragments per step (the other half of the wave's lanes' addresses: same bank pattern)
    s = s.replace('''        for (int q4 = 0; q4 < 9; ++q4) a[q4] = up[q4];
      }''', '''        for (int q4 = 0; q4 < 9; ++q4) a[q4] = up[q4];
      }
      v4f a2[9];
      {
        const v4f* up2 = reinterpret_cast<const v4f*>(ulds + ((size_t)c4 * 64 + (lane ^ 32)) * 36);
#pragma unroll
        for (int q4 = 0; q4 < 9; ++q4) a2[q4] = up2[q4];
      }
      // the streamed slice of the 32-channel form: 18 KB per step and workgroup = five 16-byte loads + LDS stores per
      // thread into the LDS left beside the resident slice, loaded one step ahead (issued in the previous step, stored
      // now), and one workgroup barrier per step
      {
        v4f* ring = reinterpret_cast<v4f*>(ulds + kUldsFloats + 256);
#pragma unroll
        for (int q = 0; q < 3; ++q) ring[(q * 256 + tid) % 960] = upre[q] + upre[(q + 3) % 5];
        const v4f* usrc = reinterpret_cast<const v4f*>(U) + (size_t)((c4 + 1) & 15) * 1280 + tid;
#pragma unroll
        for (int q = 0; q < 5; ++q) upre[q] = usrc[q * 256];
        __syncthreads();
      }''')
    assert "v4f a2[9];" in s
    s = s.replace("    auto compute = [&](int c4, const Raw& w, v4f (&accr)[36], auto first) {",
                  "    auto compute = [&](int c4, const Raw& w, v4f (&accr)[36], auto first) {\n      v4f (&upre)[5] = upre_;")
    s = s.replace("  Tile cur, nxt;\n  Raw ra, rb;", "  Tile cur, nxt;\n  Raw ra, rb;\n  v4f upre_[5];\n"
                  "  for (int q = 0; q < 5; ++q) upre_[q] = reinterpret_cast<const v4f*>(U)[q * 256 + tid];")
    assert "upre_[5]" in s
    # 2. every block of six MFMAs twice: the accumulate form again with the second fragment set
    acc_block = re.search(r'          \} else \{\n            asm volatile\(\n(.*?)\n          \}\n#undef FPSG_A', s, re.S)
    assert acc_block
    second = acc_block.group(0)
    # text of the accumulate asm statement alone
    stmt = re.search(r'            asm volatile\(\n.*?\);\n', second, re.S).group(0)
    stmt2 = stmt.replace("FPSG_A(", "FPSG_A2(")
    s = s.replace("#undef FPSG_A\n", "#define FPSG_A2(c) a2[(l0 + (c)) >> 2][(l0 + (c)) & 3]\n" + stmt2 +
                  "#undef FPSG_A2\n#undef FPSG_A\n", 1)
    # 3. the same trip count for the four waves of a workgroup (a barrier sits in the step now)
    s = s.replace("  for (; g < G; g += g_stride) {", "  for (long g4 = (long)j * 4; g4 < G; g4 += g_stride, g += g_stride) {")
    s = s.replace('''  if (g < G) {
    locate(g, cur);''', '''  if ((long)j * 4 < G) {
    locate(g, cur);''')
    s = s.replace("locate(g + g_stride < G ? g + g_stride : g, nxt);", "locate(g + g_stride, nxt);")
    # the whole 160 KB of LDS
    s = s.replace("const size_t lds_bytes = (size_t)(kUldsFloats + 3 * 64) * sizeof(float);",
                  "const size_t lds_bytes = (size_t)160 * 1024;")
    return s


def main():
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(SRC, "winograd_fused.hip")).read()
    patched = patch(src).replace('#include "fpsg_common.h"', f'#include "{SRC}/fpsg_common.h"')
    path = os.path.join(OUT, "winograd_fused_k6f72.hip")
    open(path, "w").write(patched)
    flags = ("-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize "
             "-Wall -Wextra -Wno-unused-parameter").split()
    obj = os.path.join(OUT, "winograd_fused_k6f72.o")
    subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", path, "-o", obj], check=True)
    objs = [os.path.join(SRC, "build", f) for f in sorted(os.listdir(os.path.join(SRC, "build")))
            if f.endswith(".o") and f != "winograd_fused.o"]
    so = os.path.join(OUT, "libfpsg_hip_k6f72.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, obj, *objs], check=True)
    print("built", so)


if __name__ == "__main__":
    sys.exit(main())
