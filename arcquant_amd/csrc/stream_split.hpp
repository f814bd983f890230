// Work decomposition of the persistent decode GEMM (gemm_stream.hip), free of HIP types so that
// tests/test_stream_split.py can compile it with g++ and check the invariants on the host.
//
// The repacked weight is a sequence of UNITS = tile pairs (2 KB of codes, 256 K elements of 16 weight rows), row block
// major.  A workgroup owns a contiguous range of row blocks, i.e. ONE contiguous span of units; its 16 waves cut that
// span into 16 contiguous, balanced unit ranges ("stream-K inside the workgroup"): every wave streams one contiguous
// piece of memory whatever N and K are, and no wave idles while another walks a long row block.  A wave's range may
// start and end inside row blocks; with <= 16 row blocks per workgroup it spans at most TWO row blocks (a head segment
// that continues the previous wave's row block and a segment that starts a row block), whose partial 16x16 tiles meet
// in LDS after the K loop.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ARCQ_HD __host__ __device__
#else
#define ARCQ_HD
#endif

namespace arcq {

constexpr int kStreamWaves = 16;         // waves per workgroup (1024 threads, one workgroup per CU)
constexpr int kStreamMaxRowBlocks = 16;  // per workgroup: keeps every wave's range within two row blocks
constexpr int kStreamCUs = 256;

// workgroups: one per CU while that leaves <= 16 row blocks each, more (several rounds) beyond N = 65536
inline int stream_grid(int64_t row_blocks) {
  int64_t g = row_blocks < kStreamCUs ? row_blocks : kStreamCUs;
  const int64_t need = (row_blocks + kStreamMaxRowBlocks - 1) / kStreamMaxRowBlocks;
  if (need > g) g = need;
  return (int)(g < 1 ? 1 : g);
}

// row blocks [*rb0, *rb0 + *nrb) of workgroup g of G: balanced, the first (row_blocks % G) workgroups own one more
ARCQ_HD inline void stream_wg_range(int row_blocks, int G, int g, int* rb0, int* nrb) {
  const int base = row_blocks / G, extra = row_blocks - base * G;
  *rb0 = g * base + (g < extra ? g : extra);
  *nrb = base + (g < extra ? 1 : 0);
}

// units [*u0, *u0 + *n) of wave w: start(w) = floor(w * U / W)
ARCQ_HD inline int stream_wave_start(int U, int w) { return (int)(((int64_t)w * U) / kStreamWaves); }
ARCQ_HD inline void stream_wave_range(int U, int w, int* u0, int* n) {
  *u0 = stream_wave_start(U, w);
  *n = stream_wave_start(U, w + 1) - *u0;
}

}  // namespace arcq
