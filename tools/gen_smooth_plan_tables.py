"""Regenerate the plan tables of fibergen_amd/csrc/fg_fft_smooth_plans.h from the planner (fg_fft_smooth.h):
    python tools/gen_smooth_plan_tables.py
Lengths: every n in [48, 1024] with prime factors <= 13 that is a multiple of 10 or of 16 (plus 72, 144, 216), powers of two
excluded; z tables for M = n / 2 of those.  Only plans the kernels are built for (256 / 512 threads, <= 3 passes) are listed."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "fibergen_amd", "csrc", "fg_fft_smooth_plans.h")


def smooth(n):
    for p in (2, 3, 5, 7, 11, 13):
        while n % p == 0:
            n //= p
    return n == 1


def main():
    L = [n for n in range(48, 1025) if smooth(n) and n & (n - 1) and (n % 10 == 0 or n % 16 == 0 or n in (72, 144, 216))]
    Z = sorted({n // 2 for n in L if n % 2 == 0} | {n for n in L if n <= 512})
    src = r'''
#include <initializer_list>
#include <vector>
#include <cmath>
#include <cstdio>
#define FG_HOST_EMULATION
#include "fibergen_amd/csrc/fg_fft_kernels.h"
#include "fibergen_amd/csrc/fg_fft_smooth.h"
using namespace fg::fft;
int main() {
  const int L[] = {%s}, Z[] = {%s};
  printf("S");
  for (int n : L) { SmoothPlan p; if (smooth_plan_strided(n, &p) && p.threads == 256 && p.cap == 20 && p.npass >= 2 && p.npass <= 3) printf(" X(%%d, %%d, %%d, %%d, %%d)", n, p.lines, p.fac[0], p.fac[1], p.npass > 2 ? p.fac[2] : 1); }
  printf("\nZ");
  for (int m : Z) { SmoothPlan p; if (smooth_plan_z(m, &p) && p.threads == 256 && p.cap == 20 && p.npass >= 2 && p.npass <= 3) printf(" X(%%d, %%d, %%d, %%d, %%d)", m, p.lines, p.fac[0], p.fac[1], p.npass > 2 ? p.fac[2] : 1); }
  for (int nc : {3, 1}) {
    printf("\nX%%d", nc);
    for (int n : L) {
      SmoothPlan p;
      if (!smooth_plan_xfused(n, nc, &p) || p.joint != nc || p.npass < 2 || p.npass > 3 || (p.threads != 256 && p.threads != 512)) continue;
      const int C = p.lines / nc;
      if (C != 4 && C != 8 && C != 16) continue;
      printf(" X(%%d, %%d, %%d, %%d, %%d, %%d, %%d)", n, C, p.threads, p.cap, p.fac[0], p.fac[1], p.npass > 2 ? p.fac[2] : 1);
    }
  }
  printf("\n");
}
''' % (", ".join(map(str, L)), ", ".join(map(str, Z)))
    with tempfile.TemporaryDirectory() as d:
        cpp, exe = os.path.join(d, "plans.cpp"), os.path.join(d, "plans")
        open(cpp, "w").write(src)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", ROOT, "-o", exe, cpp])
        out = subprocess.check_output([exe]).decode().splitlines()
    tables = {l.split(" ", 1)[0]: (l.split(" ", 1)[1] if " " in l else "") for l in out}

    def wrap(t, width=118):
        items = re.findall(r"X\([^)]*\)", t)
        lines, cur = [], "  "
        for it in items:
            if len(cur) + len(it) + 1 > width:
                lines.append(cur.rstrip())
                cur = "  "
            cur += it + " "
        lines.append(cur.rstrip())
        return "".join(l.ljust(width) + " \\\n" for l in lines[:-1]) + lines[-1]

    s = open(HDR).read()
    for name, key in (("FG_SMOOTH_STRIDED_PLANS", "S"), ("FG_SMOOTH_Z_PLANS", "Z"), ("FG_SMOOTH_X_PLANS", "X3"), ("FG_SMOOTH_X1_PLANS", "X1")):
        i = s.index("#define " + name + "(X)")
        j = s.index("\n\n", i)
        s = s[:i] + "#define " + name + "(X) \\\n" + wrap(tables[key]) + s[j:]
    open(HDR, "w").write(s)
    print({k: len(re.findall(r"X\(", v)) for k, v in tables.items()})


if __name__ == "__main__":
    sys.exit(main())
