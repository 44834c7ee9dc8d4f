// ubench.hip -- VALU / LDS instruction-issue microbenchmark for gfx950, used to derive the integer roofline
// that bench.py quotes (DESIGN.md "Roofline").  Build: hipcc --offload-arch=gfx950 -O2 tools/ubench.hip -o tools/ubench.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                             \
    do                                                                                       \
    {                                                                                        \
        hipError_t e = (x);                                                                  \
        if (e != hipSuccess)                                                                 \
        {                                                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                           \
            exit(1);                                                                         \
        }                                                                                    \
    } while (0)

// 8 independent chains, BODY is one instruction template using %0..%7 as in/out and %8,%9 as extra inputs
#define REP8(INS)                                                                                     \
    INS("%0") INS("%1") INS("%2") INS("%3") INS("%4") INS("%5") INS("%6") INS("%7")

#define DEF_KERNEL(NAME, INS)                                                                          \
    __global__ __launch_bounds__(256) void NAME(int * out, int iters, int x, int y)                    \
    {                                                                                                  \
        int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,         \
            a6 = a0 + 6, a7 = a0 + 7;                                                                  \
        int vx = x + threadIdx.x, vy = y - threadIdx.x;                                                \
        for (int i = 0; i < iters; ++i)                                                                \
        {                                                                                              \
            asm volatile(REP8(INS) REP8(INS) REP8(INS) REP8(INS)                                       \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                         : "v"(vx), "v"(vy));                                                          \
        }                                                                                              \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;            \
    }

#define I_ADD(R) "v_add_u32 " R ", " R ", %8\n"
#define I_MAX(R) "v_max_i32 " R ", " R ", %8\n"
#define I_MAX3(R) "v_max3_i32 " R ", " R ", %8, %9\n"
#define I_ADD3(R) "v_add3_u32 " R ", " R ", %8, %9\n"
#define I_SDWA(R) "v_add_u32_sdwa " R ", " R ", sext(%8) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n"
#define I_DPP(R) "v_mov_b32_dpp " R ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(R) "v_add_u32_dpp " R ", %8, " R " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_WSHR(R) "v_mov_b32_dpp " R ", %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_PKADD(R) "v_pk_add_i16 " R ", " R ", %8\n"
#define I_PKMAX(R) "v_pk_max_i16 " R ", " R ", %8\n"
#define I_PKADDF(R) "v_pk_add_f16 " R ", " R ", %8\n"
#define I_PKMAXF(R) "v_pk_max_f16 " R ", " R ", %8\n"
#define I_PKMAX3F(R) "v_pk_maximum3_f16 " R ", " R ", %8, %9\n"
#define I_PERM(R) "v_perm_b32 " R ", " R ", %8, %9\n"
#define I_MAX3I16(R) "v_max3_i16 " R ", " R ", %8, %9\n"
#define I_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc\n"
#define I_MADU24(R) "v_mad_u32_u24 " R ", " R ", %8, %9\n"
#define I_LSHLOR(R) "v_lshl_or_b32 " R ", " R ", 1, %9\n"
#define I_BFE(R) "v_bfe_u32 " R ", " R ", 8, 5\n"
#define I_PKMAD(R) "v_pk_mad_i16 " R ", " R ", %8, %9\n"
#define I_MAXF32(R) "v_max3_f32 " R ", " R ", %8, %9\n"

DEF_KERNEL(k_add, I_ADD)
DEF_KERNEL(k_max, I_MAX)
DEF_KERNEL(k_max3, I_MAX3)
DEF_KERNEL(k_add3, I_ADD3)
DEF_KERNEL(k_sdwa, I_SDWA)
DEF_KERNEL(k_dpp, I_DPP)
DEF_KERNEL(k_adddpp, I_ADDDPP)
DEF_KERNEL(k_wshr, I_WSHR)
DEF_KERNEL(k_pkadd, I_PKADD)
DEF_KERNEL(k_pkmax, I_PKMAX)
DEF_KERNEL(k_pkaddf, I_PKADDF)
DEF_KERNEL(k_pkmaxf, I_PKMAXF)
DEF_KERNEL(k_pkmax3f, I_PKMAX3F)
DEF_KERNEL(k_perm, I_PERM)
DEF_KERNEL(k_max3i16, I_MAX3I16)
DEF_KERNEL(k_cndmask, I_CNDMASK)
DEF_KERNEL(k_madu24, I_MADU24)
DEF_KERNEL(k_lshlor, I_LSHLOR)
DEF_KERNEL(k_bfe, I_BFE)
DEF_KERNEL(k_pkmad, I_PKMAD)
DEF_KERNEL(k_max3f32, I_MAXF32)

// LDS: ds_read_b32 with a lane-linear address pattern
__global__ __launch_bounds__(256) void k_lds_b32(int * out, int iters, int x, int y)
{
    __shared__ int buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256)
        buf[i] = i + x;
    __syncthreads();
    int acc = 0;
    int idx = threadIdx.x;
    for (int i = 0; i < iters; ++i)
    {
#pragma unroll
        for (int u = 0; u < 32; ++u)
            acc += buf[(idx + u * 64 + i) & 4095];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + y;
}

typedef void (*kern_t)(int *, int, int, int);

static void run(char const * name, kern_t k, int * d_out, int waves_per_simd, double ops_per_instr)
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    int const cus    = prop.multiProcessorCount;
    int const blocks = cus * waves_per_simd; // 256 threads = 4 waves = 1 wave per SIMD per block
    int const iters  = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, 100, 3, 5);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep)
    {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d_out, iters, 3, 5);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best)
            best = ms;
    }
    double const instrs = (double)blocks * 256.0 * iters * 32.0; // lane-instructions
    double const rate   = instrs / (best * 1e-3);
    printf("%-12s waves/SIMD=%d  %8.3f ms  %8.2f T lane-instr/s  (%6.2f lanes/clk/CU @2.4GHz)  x%.0f = %7.2f Tops/s\n", name,
           waves_per_simd, best, rate * 1e-12, rate / (cus * 2.4e9), ops_per_instr, rate * ops_per_instr * 1e-12);
}

int main()
{
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  arch=%s  CUs=%d  clock=%d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    int * d_out;
    CHECK(hipMalloc(&d_out, 256 * 64 * 256 * sizeof(int)));
    struct
    {
        char const * n;
        kern_t       k;
        double       ops;
    } tab[] = {{"v_add_u32", k_add, 1},        {"v_max_i32", k_max, 1},       {"v_max3_i32", k_max3, 2},
               {"v_add3_u32", k_add3, 2},      {"add_sdwa", k_sdwa, 1},       {"mov_dpp_rshr", k_dpp, 1},
               {"add_dpp_rshr", k_adddpp, 1},  {"mov_dpp_wshr", k_wshr, 1},   {"v_pk_add_i16", k_pkadd, 2},
               {"v_pk_max_i16", k_pkmax, 2},   {"v_pk_add_f16", k_pkaddf, 2}, {"v_pk_max_f16", k_pkmaxf, 2},
               {"pk_maximum3_f16", k_pkmax3f, 4}, {"v_perm_b32", k_perm, 1},  {"v_max3_i16", k_max3i16, 2},
               {"v_cndmask", k_cndmask, 1},    {"v_mad_u32_u24", k_madu24, 1}, {"v_lshl_or", k_lshlor, 1},
               {"v_bfe_u32", k_bfe, 1},        {"v_pk_mad_i16", k_pkmad, 2},  {"v_max3_f32", k_max3f32, 2},
               {"ds_read_b32", k_lds_b32, 1}};
    for (int w : {1, 2, 4, 8})
        for (auto & t : tab)
            run(t.n, t.k, d_out, w, t.ops);
    return 0;
}
