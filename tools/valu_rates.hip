// Issue cost of the vector instructions the shading kernels are made of, measured on the device: cycles per
// instruction and wave for independent instruction streams, with one and with two waves on every SIMD.
// build: hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o tools/valu_rates     run: tools/valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define CHECK(e)                                                       \
    do {                                                               \
        hipError_t err_ = (e);                                         \
        if (err_ != hipSuccess) {                                      \
            printf("%s: %s\n", #e, hipGetErrorString(err_));           \
            return 1;                                                  \
        }                                                              \
    } while (0)

constexpr int ITERS = 2000; // x 64 instructions

// eight independent accumulators, eight instructions each per pass of the loop
#define KERNEL_F64(name, text)                                                                                        \
    __global__ void name(double* out, uint64_t* cycles, double seed) {                                                \
        double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
               a7 = a0 + 7;                                                                                           \
        const double b = seed * 0.5, c = seed * 0.25;                                                                 \
        const uint64_t t0 = __builtin_readcyclecounter();                                                             \
        for (int i = 0; i < ITERS; i++) {                                                                             \
            asm volatile(REP8(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7))        \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c) : "s20", "s21", "vcc");                                                                           \
        }                                                                                                             \
        const uint64_t t1 = __builtin_readcyclecounter();                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                           \
        if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;                                                 \
    }
#define KERNEL_U32(name, text)                                                                                        \
    __global__ void name(double* out, uint64_t* cycles, double seed) {                                                \
        uint32_t a0 = threadIdx.x + 11, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7;                                                                                         \
        const uint32_t b = static_cast<uint32_t>(seed) | 0x9E3779B1u, c = b * 7u;                                     \
        const uint64_t t0 = __builtin_readcyclecounter();                                                             \
        for (int i = 0; i < ITERS; i++) {                                                                             \
            asm volatile(REP8(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7))        \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c) : "s20", "s21", "vcc");                                                                           \
        }                                                                                                             \
        const uint64_t t1 = __builtin_readcyclecounter();                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                           \
        if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;                                                 \
    }
#define KERNEL_U64(name, text)                                                                                        \
    __global__ void name(double* out, uint64_t* cycles, double seed) {                                                \
        uint64_t a0 = threadIdx.x + 11, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7;                                                                                         \
        const uint32_t b = static_cast<uint32_t>(seed) | 0x9E3779B1u, c = b * 7u;                                     \
        const uint64_t t0 = __builtin_readcyclecounter();                                                             \
        for (int i = 0; i < ITERS; i++) {                                                                             \
            asm volatile(REP8(text(0) text(1) text(2) text(3) text(4) text(5) text(6) text(7))        \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c) : "s20", "s21", "vcc");                                                                           \
        }                                                                                                             \
        const uint64_t t1 = __builtin_readcyclecounter();                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<double>(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);      \
        if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;                                                 \
    }

#define T_FMA(r) "v_fma_f64 %" #r ", %" #r ", %8, %9\n"
#define T_MUL(r) "v_mul_f64 %" #r ", %" #r ", %8\n"
#define T_ADD(r) "v_add_f64 %" #r ", %" #r ", %8\n"
#define T_RCP(r) "v_rcp_f64 %" #r ", %" #r "\n"
#define T_RSQ(r) "v_rsq_f64 %" #r ", %" #r "\n"
#define T_MAX(r) "v_max_f64 %" #r ", %" #r ", %8\n"
#define T_MOV64(r) "v_mov_b64 %" #r ", %8\n"
#define T_CMP(r) "v_cmp_lt_f64 vcc, %" #r ", %8\n"
#define T_LDEXP(r) "v_ldexp_f64 %" #r ", %" #r ", 1\n"
#define T_MULLO(r) "v_mul_lo_u32 %" #r ", %" #r ", %8\n"
#define T_MULHI(r) "v_mul_hi_u32 %" #r ", %" #r ", %8\n"
#define T_CNDMASK(r) "v_cndmask_b32 %" #r ", %" #r ", %8, vcc\n"
#define T_CNDMASK_FRESH(r) "v_cndmask_b32 %" #r ", %8, %9, vcc\n"
#define T_CNDMASK_SGPR(r) "v_cndmask_b32_e64 %" #r ", %" #r ", %8, s[20:21]\n"
#define T_CMP_SGPR(r) "v_cmp_lt_f64_e64 s[20:21], %" #r ", %8\n"
#define T_CMPCND(r) "v_cmp_lt_u32 vcc, %" #r ", %8\nv_cndmask_b32 %" #r ", %" #r ", %9, vcc\n"
#define T_CMPCND2(r) "v_cmp_lt_u32 vcc, %" #r ", %8\nv_cndmask_b32 %" #r ", %" #r ", %9, vcc\nv_cndmask_b32 %" #r ", %" #r ", %8, vcc\n"
#define T_CMPCND4(r) "v_cmp_lt_u32 vcc, %" #r ", %8\nv_cndmask_b32 %" #r ", %" #r ", %9, vcc\nv_cndmask_b32 %" #r ", %" #r ", %8, vcc\nv_cndmask_b32 %" #r ", %" #r ", %9, vcc\nv_cndmask_b32 %" #r ", %" #r ", %8, vcc\n"
#define T_XOR(r) "v_xor_b32 %" #r ", %" #r ", %8\n"
#define T_ADD3(r) "v_add3_u32 %" #r ", %" #r ", %8, %9\n"
#define T_MOV32(r) "v_mov_b32 %" #r ", %8\n"
#define T_READLANE(r) "v_readlane_b32 s20, %" #r ", 3\n"
#define T_WRITELANE(r) "v_writelane_b32 %" #r ", s20, 3\n"
#define T_MAD64(r) "v_mad_u64_u32 %" #r ", vcc, %8, %9, %" #r "\n"
#define T_SHR64(r) "v_lshrrev_b64 %" #r ", 3, %" #r "\n"
#define T_ADD64(r) "v_lshl_add_u64 %" #r ", %" #r ", 0, %" #r "\n"
#define T_CVT(r) "v_cvt_f64_u32 %" #r ", %8\n"

// patterns over pairs of streams (the two halves of a double select): four pairs per pass
#define KERNEL_PAIRS(name, text)                                                                                      \
    __global__ void name(double* out, uint64_t* cycles, double seed) {                                                \
        uint32_t a0 = threadIdx.x + 11, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7;                                                                                         \
        const uint32_t b = static_cast<uint32_t>(seed) | 0x9E3779B1u, c = b * 7u;                                     \
        for (int i = 0; i < ITERS; i++) {                                                                             \
            asm volatile(REP8(text(0, 1) text(2, 3) text(4, 5) text(6, 7))                                            \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                         : "v"(b), "v"(c) : "s20", "s21", "vcc");                                                     \
        }                                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                           \
    }
#define P_VCC(a, b) "v_cmp_lt_u32 vcc, %" #a ", %8\nv_cndmask_b32 %" #a ", %" #a ", %9, vcc\nv_cndmask_b32 %" #b ", %" #b ", %9, vcc\n"
#define P_SGPR(a, b) "v_cmp_lt_u32_e64 s[20:21], %" #a ", %8\nv_cndmask_b32_e64 %" #a ", %" #a ", %9, s[20:21]\nv_cndmask_b32_e64 %" #b ", %" #b ", %9, s[20:21]\n"
#define P_VCC_GAP(a, b) "v_cmp_lt_u32 vcc, %" #a ", %8\nv_xor_b32 %" #b ", %" #b ", %8\nv_cndmask_b32 %" #a ", %" #a ", %9, vcc\n"
#define P_VCC_E64(a, b) "v_cmp_lt_u32 vcc, %" #a ", %8\nv_cndmask_b32_e64 %" #a ", %" #a ", %9, vcc\nv_cndmask_b32_e64 %" #b ", %" #b ", %9, vcc\n"
#define P_VCC_COPY(a, b) "v_cmp_lt_u32 vcc, %" #a ", %8\ns_mov_b64 s[20:21], vcc\nv_cndmask_b32_e64 %" #a ", %" #a ", %9, s[20:21]\nv_cndmask_b32_e64 %" #b ", %" #b ", %9, s[20:21]\n"
#define P_THREE(a, b) "v_xor_b32 %" #a ", %" #a ", %8\nv_xor_b32 %" #b ", %" #b ", %8\nv_xor_b32 %" #a ", %" #a ", %9\n"
KERNEL_PAIRS(k_p_vcc, P_VCC)
KERNEL_PAIRS(k_p_sgpr, P_SGPR)
KERNEL_PAIRS(k_p_vcc_gap, P_VCC_GAP)
KERNEL_PAIRS(k_p_vcc_e64, P_VCC_E64)
KERNEL_PAIRS(k_p_vcc_copy, P_VCC_COPY)
KERNEL_PAIRS(k_p_three, P_THREE)
// one stream only: every instruction depends on the one before it
#define KERNEL_CHAIN(name, type, init, text)                                                               \
    __global__ void name(double* out, uint64_t* cycles, double seed) {                                    \
        type a0 = init;                                                                                   \
        const type b = static_cast<type>(seed * 0.5), c = static_cast<type>(seed * 0.25);                 \
        for (int i = 0; i < ITERS; i++) {                                                                 \
            asm volatile(REP8(text(0) text(0) text(0) text(0) text(0) text(0) text(0) text(0))            \
                         : "+v"(a0) : "v"(b), "v"(c) : "vcc");                                            \
        }                                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = static_cast<double>(a0);                             \
    }
#define C_FMA(r) "v_fma_f64 %0, %0, %1, %2\n"
#define C_MULLO(r) "v_mul_lo_u32 %0, %0, %1\n"
#define C_XOR(r) "v_xor_b32 %0, %0, %1\n"
#define C_RCP(r) "v_rcp_f64 %0, %0\n"
#define C_FMA32(r) "v_fma_f32 %0, %0, %1, %2\n"
KERNEL_CHAIN(k_c_fma, double, seed + threadIdx.x, C_FMA)
KERNEL_CHAIN(k_c_mullo, uint32_t, threadIdx.x + 11, C_MULLO)
KERNEL_CHAIN(k_c_xor, uint32_t, threadIdx.x + 11, C_XOR)
KERNEL_CHAIN(k_c_rcp, double, seed + threadIdx.x, C_RCP)
KERNEL_CHAIN(k_c_fma32, float, seed + threadIdx.x, C_FMA32)
KERNEL_F64(k_fma, T_FMA)
KERNEL_F64(k_mul, T_MUL)
KERNEL_F64(k_add, T_ADD)
KERNEL_F64(k_rcp, T_RCP)
KERNEL_F64(k_rsq, T_RSQ)
KERNEL_F64(k_max, T_MAX)
KERNEL_F64(k_mov64, T_MOV64)
KERNEL_F64(k_cmp, T_CMP)
KERNEL_F64(k_ldexp, T_LDEXP)
KERNEL_U32(k_mullo, T_MULLO)
KERNEL_U32(k_mulhi, T_MULHI)
KERNEL_U32(k_cndmask, T_CNDMASK)
KERNEL_U32(k_cndmask_fresh, T_CNDMASK_FRESH)
KERNEL_U32(k_cndmask_sgpr, T_CNDMASK_SGPR)
KERNEL_U32(k_cmpcnd, T_CMPCND)
KERNEL_F64(k_cmp_sgpr, T_CMP_SGPR)
KERNEL_U32(k_cmpcnd2, T_CMPCND2)
KERNEL_U32(k_cmpcnd4, T_CMPCND4)
KERNEL_U32(k_xor, T_XOR)
KERNEL_U32(k_add3, T_ADD3)
KERNEL_U32(k_mov32, T_MOV32)
KERNEL_U32(k_readlane, T_READLANE)
KERNEL_U32(k_writelane, T_WRITELANE)
KERNEL_U64(k_mad64, T_MAD64)
KERNEL_U64(k_shr64, T_SHR64)
KERNEL_U64(k_add64, T_ADD64)

typedef void (*Kernel)(double*, uint64_t*, double);
struct Case {
    const char* name;
    Kernel kernel;
};

int main() {
    const Case cases[] = {{"v_fma_f64", k_fma},         {"v_fma_f64 chain", k_c_fma}, {"v_rcp_f64 chain", k_c_rcp}, {"v_mul_lo_u32 chain", k_c_mullo}, {"v_xor_b32 chain", k_c_xor}, {"v_fma_f32 chain", k_c_fma32}, {"v_mul_f64", k_mul},         {"v_add_f64", k_add},
                          {"v_rcp_f64", k_rcp},         {"v_rsq_f64", k_rsq},         {"v_max_f64", k_max},
                          {"v_mov_b64", k_mov64},       {"v_cmp_lt_f64", k_cmp},      {"v_ldexp_f64", k_ldexp},
                          {"v_mul_lo_u32", k_mullo},    {"v_mul_hi_u32", k_mulhi},    {"v_cndmask_b32", k_cndmask},
                          {"cndmask vcc, fresh dst", k_cndmask_fresh}, {"cndmask sgpr mask", k_cndmask_sgpr}, {"cmp_u32+cndmask (x2)", k_cmpcnd}, {"v_cmp_f64 -> sgpr", k_cmp_sgpr}, {"cmp+2 cndmask (x3)", k_cmpcnd2}, {"cmp+4 cndmask (x5)", k_cmpcnd4}, {"v_xor_b32", k_xor},
                          {"3 xor (per triple /2)", k_p_three},
                          {"cmp vcc; cnd a; cnd b", k_p_vcc},
                          {"cmp sgpr; cnd a; cnd b", k_p_sgpr},
                          {"cmp vcc; xor; cnd a", k_p_vcc_gap},
                          {"cmp vcc; cnd_e64 vcc x2", k_p_vcc_e64},
                          {"cmp vcc; s_mov; cnd sgpr x2", k_p_vcc_copy},         {"v_add3_u32", k_add3},       {"v_mov_b32", k_mov32},
                          {"v_readlane_b32", k_readlane}, {"v_writelane_b32", k_writelane}, {"v_mad_u64_u32", k_mad64},
                          {"v_lshrrev_b64", k_shr64},   {"v_lshl_add_u64", k_add64}};
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double* out;
    uint64_t* cycles;
    CHECK(hipMalloc(&out, sizeof(double) * 1024 * cus));
    CHECK(hipMalloc(&cycles, 8));
    printf("%d compute units; ns of SIMD time per instruction (8 streams; instructions within a stream are dependent)\n", cus);
    printf("%-24s %12s %12s %12s\n", "instruction", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD");
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (const Case& c : cases) {
        // wall time of the whole launch / (instructions per wave * waves per SIMD): what one instruction costs the SIMD
        double per[3];
        int k = 0;
        for (int threads : {256, 512, 1024}) {
            hipLaunchKernelGGL(c.kernel, dim3(cus), dim3(threads), 0, 0, out, cycles, 3.0);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(c.kernel, dim3(cus), dim3(threads), 0, 0, out, cycles, 3.0);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            per[k++] = (ms * 1e6 - 3000.0) / (ITERS * 64.0) / (threads / 256); // ns; about 3 us of launch overhead
        }
        printf("%-24s %12.2f %12.2f %12.2f\n", c.name, per[0], per[1], per[2]);
    }
    return 0;
}
