// What HBM rate does this chip sustain for the STREAM SHAPES of the step's bandwidth-bound kernels?  (round 4)
//   read1 / read2 : read-only, one / two input streams (the BatchNorm statistics / backward-reduce passes)
//   copy          : 1 read + 1 write (the guide's 6.29 TB/s reference)
//   r2w1 / r3w1   : 2 / 3 reads + 1 write (BatchNorm apply / backward dx)
//   rows4         : read-only, each wave-instruction = 4 rows x 256 B, rows 147 KB apart (csrc/c1r.hip's operand loads)
//   rows2         : the same bytes as 2 rows x 512 B (gemm.hip's tile loads)
// 16 bytes per lane per access, 4 accesses in flight per lane, 2048 blocks of 256 threads, 1.2 GB per stream.
// Usage: hbm_streams [MB per stream]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NR, int NWR>
__global__ __launch_bounds__(256) void stream_k(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c,
                                                f4* __restrict__ out, float* __restrict__ sink, long n4) {
  const long stride = (long)gridDim.x * 256;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 4 * stride) {
    f4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long j = i + u * stride < n4 ? i + u * stride : i;
      v[u] = a[j];
      if (NR > 1) v[u] += b[j];
      if (NR > 2) v[u] += c[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (NWR) { if (i + u * stride < n4) __builtin_nontemporal_store(v[u], &out[i + u * stride]); }
      else acc += v[u];
    }
  }
  if (!NWR) { const float s = acc.x + acc.y + acc.z + acc.w; if (s == 12345.678f) sink[0] = s; }
}

// SEG lanes x 16 B contiguous per row, 64 / SEG rows per wave-instruction, rows `row4` f4 apart; a wave owns a column of tiles
template <int SEG, bool WR = false>
__global__ __launch_bounds__(256) void rows_k(const f4* __restrict__ a, float* __restrict__ sink, long row4, int rows, long n4,
                                               f4* __restrict__ out = nullptr) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long nwaves = (long)gridDim.x * 4;
  const long tiles_per_row = row4 / SEG;                       // column tiles of SEG f4
  const long ngroups = n4 / (row4 * rows);                     // "images"
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  constexpr int RPI = 64 / SEG;                                // rows per instruction
  for (long t = wave; t < ngroups * tiles_per_row; t += nwaves) {
    const long g = t / tiles_per_row, col = (t - g * tiles_per_row) * SEG + (lane % SEG);
    const f4* base = a + g * row4 * rows + col + (long)(lane / SEG) * row4;
    for (int r = 0; r < rows; r += 8 * RPI) {
      f4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = base[(long)(r + u * RPI) * row4];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (WR) __builtin_nontemporal_store(v[u], out + (base - a) + (long)(r + u * RPI) * row4);
        else acc += v[u];
      }
    }
  }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 12345.678f) sink[0] = s;
}

int main(int argc, char** argv) {
  const long mb = argc > 1 ? atol(argv[1]) : 1208;
  const long n4 = mb * (1l << 20) / 16;
  f4 *a, *b, *c, *o; float* sink;
  hipMalloc(&a, n4 * 16); hipMalloc(&b, n4 * 16); hipMalloc(&c, n4 * 16); hipMalloc(&o, n4 * 16); hipMalloc(&sink, 16);
  hipMemset(a, 1, n4 * 16); hipMemset(b, 1, n4 * 16); hipMemset(c, 1, n4 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, int streams, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-8s %8.3f ms  %7.0f GB/s  (%d streams of %ld MB)\n", name, best, streams * (double)n4 * 16 / best / 1e6, streams, mb);
  };
  const dim3 g(2048), t(256);
  run("read1", 1, [&] { hipLaunchKernelGGL((stream_k<1, 0>), g, t, 0, 0, a, b, c, o, sink, n4); });
  run("read2", 2, [&] { hipLaunchKernelGGL((stream_k<2, 0>), g, t, 0, 0, a, b, c, o, sink, n4); });
  run("copy", 2, [&] { hipLaunchKernelGGL((stream_k<1, 1>), g, t, 0, 0, a, b, c, o, sink, n4); });
  run("r2w1", 3, [&] { hipLaunchKernelGGL((stream_k<2, 1>), g, t, 0, 0, a, b, c, o, sink, n4); });
  run("r3w1", 4, [&] { hipLaunchKernelGGL((stream_k<3, 1>), g, t, 0, 0, a, b, c, o, sink, n4); });
  // rows: 36864-float rows (192 x 192), 256 rows per group
  const long row4 = 36864 / 4; const int rows = 256;
  run("rows4", 1, [&] { hipLaunchKernelGGL((rows_k<16>), dim3(512), t, 0, 0, a, sink, row4, rows, n4); });
  run("rows2", 1, [&] { hipLaunchKernelGGL((rows_k<32>), dim3(512), t, 0, 0, a, sink, row4, rows, n4); });
  run("rows1", 1, [&] { hipLaunchKernelGGL((rows_k<64>), dim3(512), t, 0, 0, a, sink, row4, rows, n4); });
  // image-shaped pieces (the grouped 3x3 kernels' tiles): planes of 192 x 192 / 384 x 384 floats, a wave walks down a column of
  // 128- / 256- / 512-byte row pieces; read-only and read + write of the same pieces
  run("i192r128", 1, [&] { hipLaunchKernelGGL((rows_k<8>), dim3(1024), t, 0, 0, a, sink, 48l, 192, n4, o); });
  run("i192r256", 1, [&] { hipLaunchKernelGGL((rows_k<16>), dim3(1024), t, 0, 0, a, sink, 48l, 192, n4, o); });
  run("i384r128", 1, [&] { hipLaunchKernelGGL((rows_k<8>), dim3(1024), t, 0, 0, a, sink, 96l, 384, n4, o); });
  run("i384r512", 1, [&] { hipLaunchKernelGGL((rows_k<32>), dim3(1024), t, 0, 0, a, sink, 96l, 384, n4, o); });
  run("i192c128", 2, [&] { hipLaunchKernelGGL((rows_k<8, true>), dim3(1024), t, 0, 0, a, sink, 48l, 192, n4, o); });
  run("i192c256", 2, [&] { hipLaunchKernelGGL((rows_k<16, true>), dim3(1024), t, 0, 0, a, sink, 48l, 192, n4, o); });
  run("i384c128", 2, [&] { hipLaunchKernelGGL((rows_k<8, true>), dim3(1024), t, 0, 0, a, sink, 96l, 384, n4, o); });
  run("i384c512", 2, [&] { hipLaunchKernelGGL((rows_k<32, true>), dim3(1024), t, 0, 0, a, sink, 96l, 384, n4, o); });
  return 0;
}
