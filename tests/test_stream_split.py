"""Host check of the persistent decode GEMM's work decomposition (arcquant_amd/csrc/stream_split.hpp, compiled here with
g++): for every (row blocks, pairs) the workgroup ranges tile the row blocks exactly, the 16 wave ranges tile a workgroup's
units exactly, are balanced within one unit, and -- the property the kernel's two partial-tile slots per wave rely on -- no
wave's range is longer than one row block, so it touches at most two row blocks; every row block has exactly one wave whose
range contains its first unit (its owner in the reduction)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <stdio.h>
#include <initializer_list>
#include "stream_split.hpp"
using namespace arcq;
int main() {
  long bad = 0, cases = 0;
  for (long row_blocks : {1L, 2L, 15L, 64L, 224L, 255L, 256L, 257L, 672L, 896L, 2368L, 4095L, 4096L, 4097L, 9500L})
    for (int pairs : {1, 2, 3, 5, 8, 15, 17, 33, 57, 75, 113}) {
      const int G = stream_grid(row_blocks);
      if (G < 1 || G > row_blocks) { printf("grid %d for %ld row blocks\n", G, row_blocks); ++bad; }
      int next_rb = 0;
      for (int g = 0; g < G; ++g) {
        int rb0, nrb;
        stream_wg_range((int)row_blocks, G, g, &rb0, &nrb);
        if (rb0 != next_rb || nrb < 1 || nrb > kStreamMaxRowBlocks) { printf("wg range: rb %ld G %d g %d -> %d +%d\n", row_blocks, G, g, rb0, nrb); ++bad; }
        next_rb = rb0 + nrb;
        const int U = nrb * pairs;
        int next_u = 0, owners = 0, lo = 1 << 30, hi = 0;
        for (int w = 0; w < kStreamWaves; ++w) {
          int u0, n;
          stream_wave_range(U, w, &u0, &n);
          if (u0 != next_u || n < 0) { printf("wave gap\n"); ++bad; }
          if (n > pairs) { printf("wave range %d longer than a row block (%d pairs)\n", n, pairs); ++bad; }
          next_u = u0 + n;
          lo = n < lo ? n : lo; hi = n > hi ? n : hi;
          // owners: units that start a row block inside this range
          for (int u = u0; u < u0 + n; ++u) owners += (u % pairs == 0);
          // at most two row blocks touched
          if (n > 0 && (u0 + n - 1) / pairs - u0 / pairs > 1) { printf("three row blocks in one range\n"); ++bad; }
          ++cases;
        }
        if (next_u != U) { printf("unit coverage\n"); ++bad; }
        if (hi - lo > 1) { printf("imbalance %d..%d\n", lo, hi); ++bad; }
        if (owners != nrb) { printf("owners %d != %d\n", owners, nrb); ++bad; }
      }
      if (next_rb != row_blocks) { printf("row block coverage\n"); ++bad; }
    }
  printf("%ld wave ranges, %ld bad\n", cases, bad);
  return bad != 0;
}
"""


def test_stream_split_invariants(tmp_path):
    src = tmp_path / "split_check.cpp"
    src.write_text(SRC)
    exe = tmp_path / "split_check"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "arcquant_amd", "csrc"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout
