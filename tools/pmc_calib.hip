// pmc_calib.hip -- known-size access shapes for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (GPU box only).
// Every kernel moves an exactly known number of bytes of a 2 GiB buffer (well beyond the 256 MiB Infinity Cache, so the
// traffic is HBM's) in ONE of the access shapes this library's kernels use; tools/pmc_calib.sh runs the binary under
// `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` and tools/pmc_calib.py divides reported by known bytes per shape.
//   rd16 / rd8 / rd4     coalesced grid-stride reads, 16 / 8 / 4 bytes per lane (k_pack, k_part_count ... / records / S, P)
//   rd_row128            whole 128-byte rows (8 lanes x 16 B) at pseudo-random row indices: segment reads of a partition
//   rd_rand4             4-byte reads at pseudo-random word indices (table probes of the direct forms)
//   wr16 / wr8 / wr4     coalesced writes
//   wr_row128            whole 128-byte rows at pseudo-random row indices: what the LDS stager flushes (16 records x 8 B)
//   wr_rand4             4-byte writes at pseudo-random word indices
// build: hipcc --offload-arch=gfx950 -O2 -o tools/pmc_calib tools/pmc_calib.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef uint64_t u64;
typedef uint32_t u32;

#define CHK(e)                                                                      \
  do {                                                                              \
    hipError_t r_ = (e);                                                            \
    if (r_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_));                  \
      std::exit(1);                                                                 \
    }                                                                               \
  } while (0)

__device__ __forceinline__ u64 mix(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  return x ^ (x >> 33);
}

template <typename T>
__global__ void __launch_bounds__(256) rd(const T* __restrict__ in, u64 n, u64* __restrict__ sink) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    const T v = in[i];
    const u32* w = reinterpret_cast<const u32*>(&v);
    for (unsigned t = 0; t < sizeof(T) / 4; t++) acc += w[t];
  }
  if (acc == 0x123456789abcdefULL) sink[0] = acc;  // (never true in practice: keeps the loads alive)
}

// n_rows rows of 128 bytes each, row index pseudo-random over all rows of the buffer; 8 lanes x 16 B per row
__global__ void __launch_bounds__(256) rd_row128(const uint4* __restrict__ in, u64 rows_total, u64 n_rows, u64* __restrict__ sink) {
  u64 acc = 0;
  const u64 lane8 = threadIdx.x & 7;
  for (u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 3; g < n_rows; g += ((u64)gridDim.x * blockDim.x) >> 3) {
    const u64 row = mix(g) % rows_total;
    const uint4 v = in[row * 8 + lane8];
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 0x123456789abcdefULL) sink[0] = acc;
}

__global__ void __launch_bounds__(256) rd_rand4(const u32* __restrict__ in, u64 words_total, u64 n, u64* __restrict__ sink) {
  u64 acc = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) acc += in[mix(i) % words_total];
  if (acc == 0x123456789abcdefULL) sink[0] = acc;
}

template <typename T>
__global__ void __launch_bounds__(256) wr(T* __restrict__ out, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {
    T v;
    u32* w = reinterpret_cast<u32*>(&v);
    for (unsigned t = 0; t < sizeof(T) / 4; t++) w[t] = (u32)i + t;
    out[i] = v;
  }
}

__global__ void __launch_bounds__(256) wr_row128(uint4* __restrict__ out, u64 rows_total, u64 n_rows) {
  const u64 lane8 = threadIdx.x & 7;
  for (u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 3; g < n_rows; g += ((u64)gridDim.x * blockDim.x) >> 3) {
    const u64 row = mix(g) % rows_total;
    out[row * 8 + lane8] = make_uint4((u32)g, (u32)lane8, 3u, 4u);
  }
}

__global__ void __launch_bounds__(256) wr_rand4(u32* __restrict__ out, u64 words_total, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) out[mix(i) % words_total] = (u32)i;
}

int main() {
  const u64 bytes = 2ull << 30;
  void* buf = nullptr;
  u64* sink = nullptr;
  CHK(hipMalloc(&buf, bytes));
  CHK(hipMalloc(&sink, 64));
  CHK(hipMemset(buf, 1, bytes));
  CHK(hipDeviceSynchronize());
  const dim3 grid(256 * 16), block(256);
  const u64 n_rows = bytes / 128 / 4;   // a quarter of the rows, at random
  const u64 n_rand = bytes / 4 / 64;    // one 4-byte access per 256 bytes of buffer on average
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL((rd<uint4>), grid, block, 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL((rd<u64>), grid, block, 0, 0, (const u64*)buf, bytes / 8, sink);
    hipLaunchKernelGGL((rd<u32>), grid, block, 0, 0, (const u32*)buf, bytes / 4, sink);
    hipLaunchKernelGGL(rd_row128, grid, block, 0, 0, (const uint4*)buf, bytes / 128, n_rows, sink);
    hipLaunchKernelGGL(rd_rand4, grid, block, 0, 0, (const u32*)buf, bytes / 4, n_rand, sink);
    hipLaunchKernelGGL((wr<uint4>), grid, block, 0, 0, (uint4*)buf, bytes / 16);
    hipLaunchKernelGGL((wr<u64>), grid, block, 0, 0, (u64*)buf, bytes / 8);
    hipLaunchKernelGGL((wr<u32>), grid, block, 0, 0, (u32*)buf, bytes / 4);
    hipLaunchKernelGGL(wr_row128, grid, block, 0, 0, (uint4*)buf, bytes / 128, n_rows);
    hipLaunchKernelGGL(wr_rand4, grid, block, 0, 0, (u32*)buf, bytes / 4, n_rand);
    CHK(hipGetLastError());
    CHK(hipDeviceSynchronize());
  }
  // the known byte counts, for tools/pmc_calib.py
  std::printf("{\"buffer_bytes\": %llu, \"rd<uint4>\": %llu, \"rd<u64>\": %llu, \"rd<u32>\": %llu, \"rd_row128\": %llu, \"rd_rand4\": %llu, "
              "\"wr<uint4>\": %llu, \"wr<u64>\": %llu, \"wr<u32>\": %llu, \"wr_row128\": %llu, \"wr_rand4\": %llu}\n",
              (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)bytes, (unsigned long long)bytes,
              (unsigned long long)(n_rows * 128), (unsigned long long)(n_rand * 4), (unsigned long long)bytes, (unsigned long long)bytes,
              (unsigned long long)bytes, (unsigned long long)(n_rows * 128), (unsigned long long)(n_rand * 4));
  CHK(hipFree(buf));
  CHK(hipFree(sink));
  return 0;
}
