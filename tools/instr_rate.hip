// tools/instr_rate.hip -- per-instruction issue rates on gfx950 (inline asm, independent chains),
// used to price the murmur64 instruction mix in DESIGN.md.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

// 8 independent accumulators per lane, 8 instructions per inner step
#define BODY8(INSTR)                                     \
  for (int i = 0; i < ITERS; i++) {                      \
    asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)                 \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) \
                 : "v"(b), "s"(c));                      \
  }

#define I_MUL_LO(n) "v_mul_lo_u32 %" #n ", %" #n ", %9\n"
#define I_MUL_HI(n) "v_mul_hi_u32 %" #n ", %" #n ", %9\n"
#define I_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 7\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %8\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %8\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 8\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %8\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_MUL_LO_V(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_ADD_S(n) "v_add_u32 %" #n ", %9, %" #n "\n"
#define I_ADD_E64(n) "v_add_u32_e64 %" #n ", %" #n ", %8\n"
#define I_MAD_U32_U24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %8\n"
#define I_LSHRREV(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define I_AND_OR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %8\n"
#define I_XOR_LIT(n) "v_xor_b32 %" #n ", 0x12345679, %" #n "\n"
#define I_ADD_INL(n) "v_add_u32 %" #n ", 7, %" #n "\n"
#define I_XOR_S(n) "v_xor_b32 %" #n ", %9, %" #n "\n"
#define I_SDWA(n) "v_lshlrev_b32_sdwa %" #n ", 3, %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_ALIGNBYTE(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, 1\n"
#define I_CMP(n) "v_cmp_lt_u32 vcc, %" #n ", %8\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define I_ADDC(n) "v_addc_co_u32 %" #n ", vcc, %" #n ", %8, vcc\n"
#define I_AND_LIT(n) "v_and_b32 %" #n ", 0xDFDFDFDF, %" #n "\n"

template <int OP>
__global__ __launch_bounds__(256) void k32(uint32_t* out, uint32_t b, uint32_t c) {
  uint32_t a[8];
  for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
  if (OP == 0) BODY8(I_MUL_LO)
  if (OP == 1) BODY8(I_MUL_HI)
  if (OP == 2) BODY8(I_ADD)
  if (OP == 3) BODY8(I_XOR)
  if (OP == 4) BODY8(I_ALIGNBIT)
  if (OP == 5) BODY8(I_ADD3)
  if (OP == 6) BODY8(I_LSHLADD)
  if (OP == 7) BODY8(I_BFE)
  if (OP == 8) BODY8(I_PERM)
  if (OP == 9) BODY8(I_CNDMASK)
  if (OP == 10) BODY8(I_MUL24)
  if (OP == 11) BODY8(I_MUL_LO_V)
  if (OP == 12) BODY8(I_ADD_S)
  if (OP == 15) BODY8(I_ADD_E64)
  if (OP == 16) BODY8(I_MAD_U32_U24)
  if (OP == 17) BODY8(I_LSHRREV)
  if (OP == 18) BODY8(I_AND_OR)
  if (OP == 19) BODY8(I_XOR_LIT)
  if (OP == 20) BODY8(I_ADD_INL)
  if (OP == 21) BODY8(I_XOR_S)
  if (OP == 22) BODY8(I_SDWA)
  if (OP == 23) BODY8(I_ALIGNBYTE)
  if (OP == 24) BODY8(I_CMP)
  if (OP == 25) BODY8(I_MOV)
  if (OP == 26) BODY8(I_ADDCO)
  if (OP == 27) BODY8(I_ADDC)
  if (OP == 28) BODY8(I_AND_LIT)
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// 64-bit destination forms: 4 independent 64-bit accumulators
#define BODY4_64(INSTR)                                  \
  for (int i = 0; i < ITERS; i++) {                      \
    asm volatile(INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(0) INSTR(1) INSTR(2) INSTR(3)  \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "s"(c), "v"(d)); \
  }
#define J_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %4, %5, %" #n "\n"
#define J_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 2, %6\n"
#define J_SHL64(n) "v_lshlrev_b64 %" #n ", 3, %" #n "\n"
#define J_SHR64(n) "v_lshrrev_b64 %" #n ", 3, %" #n "\n"
#define J_MAD64_V(n) "v_mad_u64_u32 %" #n ", vcc, %4, %4, %" #n "\n"
#define J_MAD64_0(n) "v_mad_u64_u32 %" #n ", vcc, %4, %5, 0\n"
#define J_MAD64_V0(n) "v_mad_u64_u32 %" #n ", vcc, %4, %4, 0\n"
#define J_MOV64(n) "v_mov_b64 %" #n ", %6\n"
#define J_LSHLADD64_0(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %6\n"

template <int OP>
__global__ __launch_bounds__(256) void k64(uint64_t* out, uint32_t b, uint32_t c, uint64_t d) {
  uint64_t a[4];
  for (int i = 0; i < 4; i++) a[i] = threadIdx.x * 7 + i;
  if (OP == 0) BODY4_64(J_MAD64)
  if (OP == 1) BODY4_64(J_LSHLADD64)
  if (OP == 2) BODY4_64(J_SHL64)
  if (OP == 3) BODY4_64(J_SHR64)
  if (OP == 4) BODY4_64(J_MAD64_V)
  if (OP == 5) BODY4_64(J_MAD64_0)
  if (OP == 6) BODY4_64(J_MAD64_V0)
  if (OP == 7) BODY4_64(J_MOV64)
  if (OP == 8) BODY4_64(J_LSHLADD64_0)
  out[blockIdx.x * blockDim.x + threadIdx.x] = a[0] ^ a[1] ^ a[2] ^ a[3];
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount, blocks = cus * 8, threads = 256;
  uint64_t* out;
  CHECK(hipMalloc(&out, (size_t)blocks * threads * 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("gfx950 instruction issue rates, %d CUs, 8 waves/SIMD, independent chains; lanes/clk/CU at the clock the chip held\n", cus);
  printf("(wall-clock based: 'lanes/clk/CU @2.4GHz' assumes 2.4 GHz; full rate would read 128 at that clock)\n");
#define RUN32(OP, NAME) { \
    hipLaunchKernelGGL(k32<OP>, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)out, 12345u, 0x9e3779b1u); CHECK(hipDeviceSynchronize()); \
    CHECK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k32<OP>, dim3(blocks), dim3(threads), 0, 0, (uint32_t*)out, 12345u + r, 0x9e3779b1u); \
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5; \
    double ops = (double)blocks * threads * ITERS * 8; \
    printf("%-18s %7.3f ms %8.2f T lane-ops/s %7.1f lanes/clk/CU @2.4GHz\n", NAME, ms, ops / ms / 1e9, ops / (ms * 1e-3) / cus / 2.4e9); }
#define RUN64(OP, NAME) { \
    hipLaunchKernelGGL(k64<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 0x9e3779b1u, 77ull); CHECK(hipDeviceSynchronize()); \
    CHECK(hipEventRecord(e0)); for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k64<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345u + r, 0x9e3779b1u, 77ull); \
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5; \
    double ops = (double)blocks * threads * ITERS * 8; \
    printf("%-18s %7.3f ms %8.2f T lane-ops/s %7.1f lanes/clk/CU @2.4GHz\n", NAME, ms, ops / ms / 1e9, ops / (ms * 1e-3) / cus / 2.4e9); }
  RUN32(2, "v_add_u32") RUN32(3, "v_xor_b32") RUN32(0, "v_mul_lo_u32") RUN32(1, "v_mul_hi_u32") RUN32(10, "v_mul_u32_u24")
  RUN32(4, "v_alignbit_b32") RUN32(5, "v_add3_u32") RUN32(6, "v_lshl_add_u32") RUN32(7, "v_bfe_u32") RUN32(8, "v_perm_b32") RUN32(9, "v_cndmask_b32")
  RUN32(11, "mul_lo v,v,v") RUN32(12, "v_add_u32 sgpr") RUN32(15, "v_add_u32_e64")
  RUN32(16, "v_mad_u32_u24") RUN32(17, "v_lshrrev_b32") RUN32(18, "v_and_or_b32")
  RUN32(19, "v_xor_b32 literal") RUN32(28, "v_and_b32 literal") RUN32(20, "v_add_u32 inline7") RUN32(21, "v_xor_b32 sgpr") RUN32(22, "v_lshlrev_sdwa")
  RUN32(23, "v_alignbyte_b32") RUN32(24, "v_cmp_lt_u32 vcc") RUN32(25, "v_mov_b32") RUN32(26, "v_add_co_u32") RUN32(27, "v_addc_co_u32")
  RUN64(0, "v_mad_u64_u32") RUN64(1, "v_lshl_add_u64") RUN64(2, "v_lshlrev_b64") RUN64(3, "v_lshrrev_b64")
  RUN64(4, "mad_u64 v,v,v64") RUN64(5, "mad_u64 v,s,0") RUN64(6, "mad_u64 v,v,0") RUN64(7, "v_mov_b64") RUN64(8, "lshl_add_u64 sh0")
  return 0;
}
