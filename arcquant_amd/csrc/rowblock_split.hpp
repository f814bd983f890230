// K split of the repacked decode GEMM (gemm_rowblock.hip), kept free of HIP types so that tests/test_rowblock_split.py
// can compile it with g++ and check the invariants on the host: every slice of a row block owns >= 1 tile pair, the
// slices tile [0, pairs) exactly, and no slice addresses a pair outside its row block (an empty trailing slice once read
// 2 KB past the end of the last row block: ADVICE r1).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ARCQ_HD __host__ __device__
#else
#define ARCQ_HD
#endif

namespace arcq {

// waves per row block: split while that still leaves <= 2048 wave tasks (half the chip's wave slots) or >= 7 tile pairs
// per wave.  Measured (tools/repacked_bench.py, us): N=37888 K=3648 S=1/2/4: 23.7 / 20.6 / 23.5; N=10752: 11.5 / 8.8 / 10.0;
// N=4096 K=4160: 10.7 / 7.8 / 6.4 / 6.1 (S=8).  Always <= pairs.
inline int rowblock_choose_slices(int64_t row_blocks, int pairs) {
  int s = 1;
  while (s < 8 && pairs / (s * 2) >= 1 && (row_blocks * s * 2 <= 2048 || pairs / (s * 2) >= 7)) s *= 2;
  return s;
}

// balanced: the first (pairs % slices) slices own one pair more
ARCQ_HD inline void rowblock_slice_range(int pairs, int slices, int slice, int* begin, int* count) {
  const int base = pairs / slices, extra = pairs - base * slices;
  *begin = slice * base + (slice < extra ? slice : extra);
  *count = base + (slice < extra ? 1 : 0);
}

// first pair a slice's loads may address: its own first pair, or pair 0 of the row block when it owns none
ARCQ_HD inline int rowblock_load_base(int begin, int count) { return count > 0 ? begin : 0; }

}  // namespace arcq
