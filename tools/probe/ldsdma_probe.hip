// probe: layout of __builtin_amdgcn_global_load_lds with 16-byte size on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(const unsigned* __restrict__ src, unsigned* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) unsigned sm[4096];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  // every lane loads 16 bytes from a lane-dependent address: chunk index (63 - lane) of this wave's 1 KiB slice
  const unsigned* g = src + wave * 256 + (63 - lane) * 4;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(sm + wave * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = t; i < 1024; i += 256) out[i] = sm[i];
}
int main() {
  std::vector<unsigned> h(1024), o(1024);
  for (int i = 0; i < 1024; ++i) h[i] = i;
  unsigned *d, *r;
  hipMalloc(&d, 4096); hipMalloc(&r, 4096);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d, r);
  hipMemcpy(o.data(), r, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 4; ++w)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const unsigned want = w * 256 + (63 - l) * 4 + j;   // LDS chunk l of wave w holds what lane l loaded
        if (o[w * 256 + l * 4 + j] != want) ++bad;
      }
  printf("ldsdma probe: %d mismatches; o[0..7] = %u %u %u %u %u %u %u %u\n", bad, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
  return bad != 0;
}
