// tools/microbench.hip -- integer-throughput probes for gfx950 used by DESIGN.md to price the
// sketch kernel: instruction rates of the multiply-class ops murmur64 is made of, and the rate of
// the bare murmur64 (31-byte message) with operands in registers = the ceiling of any k-mer
// hashing kernel on this chip.  Build: hipcc --offload-arch=gfx950 -O3 microbench.hip -o microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

template <int OP>
__global__ __launch_bounds__(256) void k_op(uint32_t* out, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x1234567, a3 = a0 + 77;
  uint64_t b0 = ((uint64_t)a0 << 32) | a1, b1 = ((uint64_t)a2 << 32) | a3, b2 = b0 ^ 0x9e3779b97f4a7c15ull, b3 = b1 + 12345;
  for (int i = 0; i < ITERS; i++) {
    if (OP == 0) { a0 = a0 * 0x9e3779b1u + 1; a1 = a1 * 0x85ebca6bu + 1; a2 = a2 * 0xc2b2ae35u + 1; a3 = a3 * 0x27d4eb2fu + 1; }          // v_mul_lo_u32 (+add)
    if (OP == 1) { a0 = __umulhi(a0, 0x9e3779b1u) + 1; a1 = __umulhi(a1, 0x85ebca6bu) + 1; a2 = __umulhi(a2, 0xc2b2ae35u) + 1; a3 = __umulhi(a3, 0x27d4eb2fu) + 1; }
    if (OP == 2) { b0 = (uint64_t)(uint32_t)b0 * 0x9e3779b1u + b0; b1 = (uint64_t)(uint32_t)b1 * 0x85ebca6bu + b1; b2 = (uint64_t)(uint32_t)b2 * 0xc2b2ae35u + b2; b3 = (uint64_t)(uint32_t)b3 * 0x27d4eb2fu + b3; }  // v_mad_u64_u32
    if (OP == 3) { b0 *= 0x87c37b91114253d5ull; b1 *= 0x4cf5ad432745937full; b2 *= 0xff51afd7ed558ccdull; b3 *= 0xc4ceb9fe1a85ec53ull; }  // 64x64 low
    if (OP == 4) { a0 = (a0 + 0x9e3779b1u) ^ a1; a1 = (a1 + 0x85ebca6bu) ^ a2; a2 = (a2 + 0xc2b2ae35u) ^ a3; a3 = (a3 + 0x27d4eb2fu) ^ a0; }  // add+xor (full rate)
    if (OP == 5) { b0 = (b0 << 2) | (b0 >> 62); b1 = (b1 << 5) | (b1 >> 59); b2 = (b2 << 7) | (b2 >> 57); b3 = (b3 << 9) | (b3 >> 55); b0 += b1; b2 += b3; }  // 64-bit rotates + adds
    if (OP == 6) { b0 = (b0 << 2) + 3; b1 = (b1 >> 2) ^ b0; b2 = (b2 << 2) + 1; b3 = (b3 >> 2) ^ b2; }  // v_lshlrev_b64 / v_lshrrev_b64
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3) ^ (uint32_t)((b0 ^ b1 ^ b2 ^ b3) >> 32);
}

__device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ uint64_t fmix64(uint64_t k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return k; }
__device__ __forceinline__ uint64_t murmur31(uint64_t w0, uint64_t w1, uint64_t w2, uint64_t w3, uint64_t seed) {
  const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;
  uint64_t h1 = seed, h2 = seed, k1 = w0, k2 = w1;
  k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
  k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
  k2 = w3; k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
  k1 = w2; k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
  h1 ^= 31; h2 ^= 31; h1 += h2; h2 += h1; h1 = fmix64(h1); h2 = fmix64(h2);
  return h1 + h2;
}

// bare murmur64 of 31-byte messages held in registers: ILP = 2 independent messages per lane
__global__ __launch_bounds__(256) void k_murmur(uint64_t* out, uint64_t seed) {
  uint64_t w0 = threadIdx.x * 0x9e3779b97f4a7c15ull + blockIdx.x, w1 = w0 ^ 0x1111, w2 = w0 + 99, w3 = (w0 >> 8);
  uint64_t acc = 0;
  for (int i = 0; i < ITERS / 8; i++) {
    uint64_t h = murmur31(w0, w1, w2, w3, seed);
    uint64_t g = murmur31(w1, w2, w3 ^ i, w0, seed);
    acc ^= h ^ g;
    w0 += h; w1 ^= g; w2 += 0x41434754; w3 = (w3 + h) >> 8;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int blocks = cus * 8, threads = 256;
  uint32_t* out;
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  const char* names[] = {"v_mul_lo_u32(+add)", "v_mul_hi_u32(+add)", "v_mad_u64_u32", "mul64x64lo (C++)", "add+xor 32-bit", "rotl64+add64 (C++)", "shl/shr 64 (C++)"};
  const double ops_per_iter[] = {4, 4, 4, 4, 8, 6, 6};
  printf("device %s, %d CUs, clock %d MHz\n", p.name, cus, p.clockRate / 1000);
#define RUN(OP) { \
    hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1u); \
    CHECK(hipDeviceSynchronize()); \
    CHECK(hipEventRecord(a)); \
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_op<OP>, dim3(blocks), dim3(threads), 0, 0, out, (uint32_t)r); \
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); \
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 5; \
    double ops = (double)blocks * threads * ITERS * ops_per_iter[OP]; \
    printf("%-22s %8.3f ms  %8.2f T lane-ops/s  (%.1f lane-ops/clk/CU at 2.4 GHz)\n", names[OP], ms, ops / ms / 1e9, ops / (ms * 1e-3) / cus / 2.4e9); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
  {
    hipLaunchKernelGGL(k_murmur, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)out, 42ull);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_murmur, dim3(blocks), dim3(threads), 0, 0, (uint64_t*)out, 42ull + r);
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    double hashes = (double)blocks * threads * (ITERS / 8) * 2;
    printf("bare murmur64(31 B)    %8.3f ms  %8.2f G hashes/s   <- ceiling for k=31 k-mers/s on this chip\n", ms, hashes / ms / 1e6);
  }
  return 0;
}
