"""Host check of the repacked decode GEMM's K split (arcquant_amd/csrc/rowblock_split.hpp, compiled here with g++):
for every K the path supports, the chosen slices are all non-empty, tile the row block's pairs exactly, and never
address a pair outside the row block -- the out-of-bounds read of ADVICE r1 (pairs=17, 8 slices: slices 6 and 7 were
empty and read 2 KB past the last row block) cannot come back unnoticed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <stdio.h>
#include <initializer_list>
#include "rowblock_split.hpp"
int main() {
  long bad = 0, cases = 0;
  for (long row_blocks : {1L, 64L, 224L, 256L, 896L, 2368L, 4096L})
    for (int pairs = 1; pairs <= 160; ++pairs)
      for (int forced = 0; forced <= 8; forced = forced ? forced * 2 : 1) {
        const int s = forced ? forced : arcq::rowblock_choose_slices(row_blocks, pairs);
        if (!forced && s > pairs) { printf("slices %d > pairs %d\n", s, pairs); ++bad; }
        int next = 0;
        for (int slice = 0; slice < s; ++slice) {
          int b, n;
          arcq::rowblock_slice_range(pairs, s, slice, &b, &n);
          if (b != next || n < 0) { printf("gap: pairs %d s %d slice %d\n", pairs, s, slice); ++bad; }
          if (!forced && n < 1) { printf("empty slice: pairs %d s %d slice %d\n", pairs, s, slice); ++bad; }
          next = b + n;
          const int lb = arcq::rowblock_load_base(b, n);
          // the kernel keeps three loads in flight, each clamped to the slice's last pair
          const int last = lb + (n > 0 ? n - 1 : 0);
          if (lb < 0 || last >= pairs) { printf("oob: pairs %d s %d slice %d\n", pairs, s, slice); ++bad; }
          ++cases;
        }
        if (next != pairs) { printf("coverage: pairs %d s %d\n", pairs, s); ++bad; }
      }
  // the shape ADVICE r1 names: N=4096, K=4160 -> pairs=17, 8 slices
  if (arcq::rowblock_choose_slices(256, 17) != 8) { printf("config[1] no longer takes 8 slices\n"); ++bad; }
  printf("%ld cases, %ld bad\n", cases, bad);
  return bad != 0;
}
"""


def test_rowblock_split_invariants(tmp_path):
    src = tmp_path / "split_check.cpp"
    src.write_text(SRC)
    exe = tmp_path / "split_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "arcquant_amd", "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout
