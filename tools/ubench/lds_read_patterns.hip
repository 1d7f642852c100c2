// LDS read throughput of the fragment-read address patterns the kernels use, gfx950: cycles per wave-instruction with 1, 2 and 4 waves
// per SIMD all reading (one workgroup per CU), for ds_read_b128 and ds_read_b64_tr_b16.  The patterns are (lane -> byte address) tables
// built on the host; the kernel only replays them, so a new layout is one more table.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/lds_read_patterns.hip -o tools/ubench/lds_read_patterns.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
#include <functional>

#define REPS 64
#define UNROLL 16

template <int KIND>   // 0: ds_read_b128, 1: ds_read_b64_tr_b16, 2: ds_read_b64
__global__ __launch_bounds__(1024) void k_read(const uint32_t* __restrict__ addr_tab, uint64_t* out, int stride_per_wave) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 160 * 1024 / 16 - 64; i += blockDim.x) reinterpret_cast<uint4*>(smem)[i] = uint4{1u, 2u, 3u, 4u};
  __syncthreads();
  const uint32_t a = addr_tab[lane] + (uint32_t)(wave * stride_per_wave);
  uint32_t acc = 0;
  __builtin_amdgcn_s_barrier();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int r = 0; r < REPS; ++r) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      if constexpr (KIND == 0) {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a) : "memory");
        asm volatile("" :: "v"(v));
      } else if constexpr (KIND == 1) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 v;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(a) : "memory");
        asm volatile("" :: "v"(v));
      } else {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        u32x2 v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a) : "memory");
        asm volatile("" :: "v"(v));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 16 + wave] = t1 - t0;
  if (acc == 0x12345678u) out[4096] = acc;
}

struct Pattern { std::string name; int kind; std::function<uint32_t(int)> addr; };

int main() {
  uint32_t* d_tab; uint64_t* d_out;
  hipMalloc(&d_tab, 64 * 4); hipMalloc(&d_out, 8192 * 8);
  std::vector<Pattern> ps;
  // ---- ds_read_b128 (16 B per lane): lane = (r = lane & 15, g = lane >> 4)
  ps.push_back({"b128 contiguous (lane * 16)", 0, [](int l) { return (uint32_t)(l * 16); }});
  ps.push_back({"b128 ff_pair image: row r, slot g ^ (row & 7), 128-B rows", 0, [](int l) { int r = l & 15, g = l >> 4; return (uint32_t)(r * 128 + ((g ^ (r & 7)) << 4)); }});
  ps.push_back({"b128 same, slot g ^ ((row >> 1) & 7)", 0, [](int l) { int r = l & 15, g = l >> 4; return (uint32_t)(r * 128 + ((g ^ ((r >> 1) & 7)) << 4)); }});
  ps.push_back({"b128 128-B rows, no swizzle (row r, slot g)", 0, [](int l) { int r = l & 15, g = l >> 4; return (uint32_t)(r * 128 + (g << 4)); }});
  ps.push_back({"b128 attention-style: row r, 16 slots, slot (4 ks + g) ^ (row & 15), 256-B rows, ks = 0", 0, [](int l) { int r = l & 15, g = l >> 4; return (uint32_t)(r * 256 + ((g ^ (r & 15)) << 4)); }});
  ps.push_back({"b128 512-B rows: row r, slot g ^ (row & 15)", 0, [](int l) { int r = l & 15, g = l >> 4; return (uint32_t)(r * 512 + ((g ^ (r & 15)) << 4)); }});
  ps.push_back({"b128 padded rows 144 B (row r, slot g)", 0, [](int l) { int r = l & 15, g = l >> 4; return (uint32_t)(r * 144 + (g << 4)); }});
  ps.push_back({"b128 padded rows 272 B (row r, slot g)", 0, [](int l) { int r = l & 15, g = l >> 4; return (uint32_t)(r * 272 + (g << 4)); }});
  // ---- ds_read_b64_tr_b16 (8 B per lane): lane = (g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3): row 8 g + q, 4 channels at 4 p
  ps.push_back({"tr16 padded 288-B rows (old wgrad tile)", 1, [](int l) { int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (uint32_t)((8 * g + q) * 288 + 8 * p); }});
  ps.push_back({"tr16 256-B rows, unit ^ (q << 2)", 1, [](int l) { int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (uint32_t)((8 * g + q) * 256 + ((((p >> 1)) ^ (q << 2)) << 4) + (p & 1) * 8); }});
  ps.push_back({"tr16 256-B rows, unit ^ (q << 2) ^ ((g & 1) << 1)", 1, [](int l) { int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (uint32_t)((8 * g + q) * 256 + ((((p >> 1)) ^ (q << 2) ^ ((g & 1) << 1)) << 4) + (p & 1) * 8); }});
  ps.push_back({"tr16 256-B rows, unit ^ (q << 2) ^ (g << 1) (4 groups apart)", 1, [](int l) { int g = l >> 4, q = (l & 15) >> 2, p = l & 3; return (uint32_t)((8 * g + q) * 256 + ((((p >> 1)) ^ (q << 2) ^ ((g & 1) << 1) ^ ((g >> 1) << 3)) << 4) + (p & 1) * 8); }});
  ps.push_back({"tr16 contiguous (lane * 8)", 1, [](int l) { return (uint32_t)(l * 8); }});
  ps.push_back({"b64 contiguous (lane * 8)", 2, [](int l) { return (uint32_t)(l * 8); }});
  for (auto& p : ps) {
    uint32_t tab[64];
    for (int l = 0; l < 64; ++l) tab[l] = p.addr(l);
    hipMemcpy(d_tab, tab, sizeof(tab), hipMemcpyHostToDevice);
    printf("%-96s", p.name.c_str());
    for (int waves : {4, 8, 16}) {
      hipMemset(d_out, 0, 8192 * 8);
      void (*k)(const uint32_t*, uint64_t*, int) = p.kind == 0 ? k_read<0> : (p.kind == 1 ? k_read<1> : k_read<2>);
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(k, dim3(256), dim3(waves * 64), 160 * 1024 - 64, 0, d_tab, d_out, 8192);
      hipDeviceSynchronize();
      std::vector<uint64_t> h(256 * 16);
      hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
      double s = 0; int n = 0;
      for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { s += h[b * 16 + w]; ++n; }
      const double per_wave_instr = s / n / (REPS * UNROLL);
      const int bytes = p.kind == 0 ? 1024 : 512;
      printf("  %2d waves/CU: %6.1f clk/instr/wave = %5.1f B/clk/CU", waves, per_wave_instr, bytes * waves / per_wave_instr);
    }
    printf("\n");
  }
  return 0;
}
