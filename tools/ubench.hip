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


#define I_SUB(R) "v_sub_u32 " R ", " R ", %8\n"
#define I_AND(R) "v_and_b32 " R ", " R ", %8\n"
#define I_OR(R) "v_or_b32 " R ", " R ", %8\n"
#define I_XOR(R) "v_xor_b32 " R ", " R ", %8\n"
#define I_MINI(R) "v_min_i32 " R ", " R ", %8\n"
#define I_MAXU(R) "v_max_u32 " R ", " R ", %8\n"
#define I_LSHL(R) "v_lshlrev_b32 " R ", 1, " R "\n"
#define I_MOV(R) "v_mov_b32 " R ", %8\n"
#define I_ADDF(R) "v_add_f32 " R ", " R ", %8\n"
#define I_MAXF(R) "v_max_f32 " R ", " R ", %8\n"
#define I_MULF(R) "v_mul_f32 " R ", " R ", %8\n"
#define I_FMAF(R) "v_fma_f32 " R ", " R ", %8, %9\n"
#define I_FMACF(R) "v_fmac_f32 " R ", %8, %9\n"
#define I_ADDE64(R) "v_add_u32_e64 " R ", " R ", %8\n"
#define I_ADDU16(R) "v_add_u16 " R ", " R ", %8\n"
#define I_MAXI16(R) "v_max_i16 " R ", " R ", %8\n"
#define I_ADDF16(R) "v_add_f16 " R ", " R ", %8\n"
#define I_MAXF16(R) "v_max_f16 " R ", " R ", %8\n"
#define I_CVTUB(R) "v_cvt_f32_ubyte1 " R ", " R "\n"
#define I_CVTI(R) "v_cvt_f32_i32 " R ", " R "\n"
#define I_MED3(R) "v_med3_i32 " R ", " R ", %8, %9\n"
#define I_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 1, %9\n"
#define I_ADDLSHL(R) "v_add_lshl_u32 " R ", " R ", %8, 1\n"
#define I_ANDOR(R) "v_and_or_b32 " R ", " R ", %8, %9\n"
#define I_MIX1(R) "v_add_u32 " R ", " R ", %8\nv_max3_i32 " R ", " R ", %8, %9\n"
#define I_MIX2(R) "v_add_u32 " R ", " R ", %8\nv_add_u32 " R ", " R ", %9\nv_max3_i32 " R ", " R ", %8, %9\nv_max_i32 " R ", " R ", %8\n"
#define I_ADDS(R) "v_add_u32 " R ", s4, " R "\n"
#define I_MAXS(R) "v_max_i32 " R ", s4, " R "\n"
#define I_SUBREV(R) "v_subrev_u32 " R ", %8, " R "\n"
#define I_ADDCO(R) "v_add_co_u32 " R ", vcc, " R ", %8\n"
#define I_MAX3F16(R) "v_max3_f16 " R ", " R ", %8, %9\n"
#define I_PKMIN(R) "v_pk_min_i16 " R ", " R ", %8\n"
#define I_PKADDU(R) "v_pk_add_u16 " R ", " R ", %8\n"
#define I_SAD(R) "v_sad_u32 " R ", " R ", %8, %9\n"
#define I_ADDFC(R) "v_add_f32 " R ", 1.0, " R "\n"

DEF_KERNEL(k_sub, I_SUB) DEF_KERNEL(k_and, I_AND) DEF_KERNEL(k_or, I_OR) DEF_KERNEL(k_xor, I_XOR)
DEF_KERNEL(k_mini, I_MINI) DEF_KERNEL(k_maxu, I_MAXU) DEF_KERNEL(k_lshl, I_LSHL) DEF_KERNEL(k_mov, I_MOV)
DEF_KERNEL(k_addf, I_ADDF) DEF_KERNEL(k_maxf, I_MAXF) DEF_KERNEL(k_mulf, I_MULF) DEF_KERNEL(k_fmaf, I_FMAF)
DEF_KERNEL(k_fmacf, I_FMACF) DEF_KERNEL(k_adde64, I_ADDE64) DEF_KERNEL(k_addu16, I_ADDU16)
DEF_KERNEL(k_maxi16, I_MAXI16) DEF_KERNEL(k_addf16, I_ADDF16) DEF_KERNEL(k_maxf16, I_MAXF16)
DEF_KERNEL(k_cvtub, I_CVTUB) DEF_KERNEL(k_cvti, I_CVTI) DEF_KERNEL(k_med3, I_MED3) DEF_KERNEL(k_lshladd, I_LSHLADD)
DEF_KERNEL(k_addlshl, I_ADDLSHL) DEF_KERNEL(k_andor, I_ANDOR) DEF_KERNEL(k_mix1, I_MIX1) DEF_KERNEL(k_mix2, I_MIX2)
DEF_KERNEL(k_adds, I_ADDS) DEF_KERNEL(k_maxs, I_MAXS) DEF_KERNEL(k_subrev, I_SUBREV) DEF_KERNEL(k_addco, I_ADDCO)
DEF_KERNEL(k_max3f16, I_MAX3F16) DEF_KERNEL(k_pkmin, I_PKMIN) DEF_KERNEL(k_pkaddu, I_PKADDU) DEF_KERNEL(k_sad, I_SAD)
DEF_KERNEL(k_addfc, I_ADDFC)
#define I_CNDS(R) "v_cndmask_b32_e64 " R ", " R ", %8, s[6:7]\n"
#define I_BFI(R) "v_bfi_b32 " R ", %8, " R ", %9\n"
#define I_PKADDSEL(R) "v_pk_add_f16 " R ", " R ", %8 op_sel_hi:[1,0]\n"
#define I_PKFMA(R) "v_pk_fma_f16 " R ", " R ", %8, %9\n"
#define I_PKADDF32(R) "v_pk_add_f32 v[20:21], v[20:21], v[22:23]\n"
// the packed sweep's own mix per pair of cells: 3 packed adds, 3 packed max3 (the two-operand maximum is encoded as one too),
// 1 byte permute -- lx_score_f16.hip / lx_score_i16.hip / lx_sweep_mq.hip
#define I_SWEEPMIX(R)                                                                                                       \
    "v_pk_add_u16 " R ", " R ", %8\n v_pk_maximum3_f16 " R ", " R ", %8, %9\n v_pk_add_u16 " R ", " R ", %9\n"               \
    "v_pk_maximum3_f16 " R ", " R ", %9, %8\n v_pk_maximum3_f16 " R ", " R ", %8, %8\n v_pk_add_u16 " R ", " R ", %8\n"       \
    "v_perm_b32 " R ", " R ", %8, %9\n"
DEF_KERNEL(k_sweepmix, I_SWEEPMIX)
// round 6: the diagonal operand's two halves added by two NON-packed half adds (the second one writes the high half: VOP3 op_sel, one wait
// state before its result is read) instead of v_perm_b32 + v_pk_add -- are they full-rate like the VOP2 forms?
// (gfx950 takes op_sel on the VOP3-only 16-bit forms: v_fma_f16, v_mad_u16, v_fma_mixhi_f16 ... not on v_add_f16)
#define I_ADDF16HI(R) "v_fma_f16 " R ", " R ", 1.0, %8 op_sel:[1,0,1,1]\n"
#define I_ADDF16SEL(R) "v_fma_f16 " R ", " R ", 1.0, %8\n"
#define I_MADU16HI(R) "v_mad_u16 " R ", " R ", 1, %8 op_sel:[1,0,1,1]\n"
#define I_MIXHI(R) "v_fma_mixhi_f16 " R ", " R ", 1.0, %8 op_sel:[1,0,1] op_sel_hi:[1,0,1]\n"
#define I_MIXLO(R) "v_fma_mixlo_f16 " R ", " R ", 1.0, %8 op_sel_hi:[1,0,1]\n"
#define I_SWEEPMIX2(R)                                                                                                      \
    "v_add_f16 " R ", " R ", %8\n v_fma_f16 " R ", " R ", 1.0, %9 op_sel:[1,0,1,1]\n s_nop 0\n v_pk_maximum3_f16 " R ", " R ", %8, %9\n v_pk_add_u16 " R ", " R ", %9\n" \
    "v_pk_maximum3_f16 " R ", " R ", %9, %8\n v_pk_maximum3_f16 " R ", " R ", %8, %8\n v_pk_add_u16 " R ", " R ", %8\n"
DEF_KERNEL(k_addf16hi, I_ADDF16HI) DEF_KERNEL(k_addf16sel, I_ADDF16SEL) DEF_KERNEL(k_sweepmix2, I_SWEEPMIX2)
DEF_KERNEL(k_madu16hi, I_MADU16HI) DEF_KERNEL(k_mixhi, I_MIXHI) DEF_KERNEL(k_mixlo, I_MIXLO)
DEF_KERNEL(k_cnds, I_CNDS) DEF_KERNEL(k_bfi, I_BFI) DEF_KERNEL(k_pkaddsel, I_PKADDSEL) DEF_KERNEL(k_pkfma, I_PKFMA)

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
    double const instrs = (double)blocks * 256.0 * iters * 32.0; // lane-instructions (mix kernels: x2 / x4 more, see name)
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
               {"ds_read_b32", k_lds_b32, 1},
               {"v_sub_u32", k_sub, 1}, {"v_and_b32", k_and, 1}, {"v_or_b32", k_or, 1}, {"v_xor_b32", k_xor, 1},
               {"v_min_i32", k_mini, 1}, {"v_max_u32", k_maxu, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_mov_b32", k_mov, 1},
               {"v_add_f32", k_addf, 1}, {"v_max_f32", k_maxf, 1}, {"v_mul_f32", k_mulf, 1}, {"v_fma_f32", k_fmaf, 2},
               {"v_fmac_f32", k_fmacf, 2}, {"v_add_u32_e64", k_adde64, 1}, {"v_add_u16", k_addu16, 1},
               {"v_max_i16", k_maxi16, 1}, {"v_add_f16", k_addf16, 1}, {"v_max_f16", k_maxf16, 1},
               {"v_cvt_f32_ubyte1", k_cvtub, 1}, {"v_cvt_f32_i32", k_cvti, 1}, {"v_med3_i32", k_med3, 2},
               {"v_lshl_add_u32", k_lshladd, 2}, {"v_add_lshl_u32", k_addlshl, 2}, {"v_and_or_b32", k_andor, 2},
               {"mix_add+max3", k_mix1, 1.5}, {"mix_2add+max3+max", k_mix2, 1.25},
               {"v_add_u32_sgpr", k_adds, 1}, {"v_max_i32_sgpr", k_maxs, 1}, {"v_subrev_u32", k_subrev, 1},
               {"v_add_co_u32", k_addco, 1}, {"v_max3_f16", k_max3f16, 2}, {"v_pk_min_i16", k_pkmin, 2},
               {"v_pk_add_u16", k_pkaddu, 2}, {"v_sad_u32", k_sad, 2}, {"v_add_f32_const", k_addfc, 1},
               {"v_cndmask_e64_sgpr", k_cnds, 1}, {"v_bfi_b32", k_bfi, 1}, {"pk_add_f16_opsel", k_pkaddsel, 2},
               {"v_pk_fma_f16", k_pkfma, 4}, {"v_fma_f16_hi(op_sel)", k_addf16hi, 1}, {"v_fma_f16", k_addf16sel, 1}, {"v_mad_u16_hi(op_sel)", k_madu16hi, 1},
               {"v_fma_mixhi_f16", k_mixhi, 1}, {"v_fma_mixlo_f16", k_mixlo, 1}};
    for (int w : {8})
        for (auto & t : tab)
            run(t.n, t.k, d_out, w, t.ops);
    // what the DP kernels are priced against: the packed classes and the sweep's own instruction mix at the occupancies the
    // sweeps run at (2 and 3 wavefronts per SIMD); "sweep_mix" executes 7 instructions per template: multiply its rate by 7
    printf("---- occupancy of the sweeps (sweep_mix: x 7 lane-instructions per counted one)\n");
    for (int w : {2, 3, 4, 8})
    {
        run("v_pk_add_u16", k_pkaddu, d_out, w, 2);
        run("pk_maximum3_f16", k_pkmax3f, d_out, w, 4);
        run("v_perm_b32", k_perm, d_out, w, 1);
        run("sweep_mix(x7)", k_sweepmix, d_out, w, 7);
        run("sweep_mix2(x7)", k_sweepmix2, d_out, w, 7);
    }
    return 0;
}
