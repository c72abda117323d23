// micro-benchmark: issue cost of individual VALU instructions on gfx950, relative to v_fma_f64 (tuning aid, not part of
// the product).  8 waves/SIMD, 4 independent chains per lane, 4096 x 16 instructions per chain per lane.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

#define KERNEL_D(name, ASM)                                                                        \
    __global__ void __launch_bounds__(256) name(double* out, int iters) {                          \
        double a = threadIdx.x * 1e-3 + 1.5, b = a + 1.0, c = a + 2.0, d = a + 3.0;                \
        int ia = threadIdx.x, ib = ia + 1, ic = ia + 2, id = ia + 3;                               \
        float fa = a, fb = b, fc = c, fd = d;                                                      \
        for (int i = 0; i < iters; ++i) {                                                          \
            REP4(asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id), \
                                    "+v"(fa), "+v"(fb), "+v"(fc), "+v"(fd));)                      \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + ia + ib + ic + id + fa + fb + fc + fd; \
    }
// operands: %0-%3 doubles, %4-%7 ints, %8-%11 floats
#define FOUR(op) op(0) op(1) op(2) op(3)

KERNEL_D(k_fma64,  "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1\n")
KERNEL_D(k_fma64_1chain, "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n")
KERNEL_D(k_fma64_2chain, "v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n")
KERNEL_D(k_add64,  "v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n")
KERNEL_D(k_mul64,  "v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0\n")
KERNEL_D(k_ldexp,  "v_ldexp_f64 %0, %0, %4\n v_ldexp_f64 %1, %1, %5\n v_ldexp_f64 %2, %2, %6\n v_ldexp_f64 %3, %3, %7\n")
KERNEL_D(k_rcp64,  "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n")
KERNEL_D(k_rsq64,  "v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n")
KERNEL_D(k_sqrt64, "v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n")
KERNEL_D(k_frexpm, "v_frexp_mant_f64 %0, %0\n v_frexp_mant_f64 %1, %1\n v_frexp_mant_f64 %2, %2\n v_frexp_mant_f64 %3, %3\n")
KERNEL_D(k_frexpe, "v_frexp_exp_i32_f64 %4, %0\n v_frexp_exp_i32_f64 %5, %1\n v_frexp_exp_i32_f64 %6, %2\n v_frexp_exp_i32_f64 %7, %3\n")
KERNEL_D(k_cvtdi,  "v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %5\n v_cvt_f64_i32 %2, %6\n v_cvt_f64_i32 %3, %7\n")
KERNEL_D(k_cvtid,  "v_cvt_i32_f64 %4, %0\n v_cvt_i32_f64 %5, %1\n v_cvt_i32_f64 %6, %2\n v_cvt_i32_f64 %7, %3\n")
KERNEL_D(k_cvtfd,  "v_cvt_f32_f64 %8, %0\n v_cvt_f32_f64 %9, %1\n v_cvt_f32_f64 %10, %2\n v_cvt_f32_f64 %11, %3\n")
KERNEL_D(k_cvtdf,  "v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %9\n v_cvt_f64_f32 %2, %10\n v_cvt_f64_f32 %3, %11\n")
KERNEL_D(k_rcp32,  "v_rcp_f32 %8, %8\n v_rcp_f32 %9, %9\n v_rcp_f32 %10, %10\n v_rcp_f32 %11, %11\n")
KERNEL_D(k_log32,  "v_log_f32 %8, %8\n v_log_f32 %9, %9\n v_log_f32 %10, %10\n v_log_f32 %11, %11\n")
KERNEL_D(k_exp32,  "v_exp_f32 %8, %8\n v_exp_f32 %9, %9\n v_exp_f32 %10, %10\n v_exp_f32 %11, %11\n")
KERNEL_D(k_fma32,  "v_fma_f32 %8, %8, %9, %10\n v_fma_f32 %9, %9, %10, %11\n v_fma_f32 %10, %10, %11, %8\n v_fma_f32 %11, %11, %8, %9\n")
KERNEL_D(k_mullo,  "v_mul_lo_u32 %4, %4, %5\n v_mul_lo_u32 %5, %5, %6\n v_mul_lo_u32 %6, %6, %7\n v_mul_lo_u32 %7, %7, %4\n")
KERNEL_D(k_mulhi,  "v_mul_hi_u32 %4, %4, %5\n v_mul_hi_u32 %5, %5, %6\n v_mul_hi_u32 %6, %6, %7\n v_mul_hi_u32 %7, %7, %4\n")
KERNEL_D(k_mad64,  "v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3\n")
KERNEL_D(k_mul24,  "v_mul_u32_u24 %4, %4, %5\n v_mul_u32_u24 %5, %5, %6\n v_mul_u32_u24 %6, %6, %7\n v_mul_u32_u24 %7, %7, %4\n")
KERNEL_D(k_lshl64, "v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 3, %1\n v_lshlrev_b64 %2, 3, %2\n v_lshlrev_b64 %3, 3, %3\n")
KERNEL_D(k_and32,  "v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %4\n")
KERNEL_D(k_mov64,  "v_mov_b64 %0, %1\n v_mov_b64 %1, %2\n v_mov_b64 %2, %3\n v_mov_b64 %3, %0\n")
KERNEL_D(k_cmp64,  "v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_lt_f64 vcc, %2, %3\n v_cndmask_b32 %6, %6, %7, vcc\n")
KERNEL_D(k_min64,  "v_min_f64 %0, %0, %1\n v_min_f64 %1, %1, %2\n v_max_f64 %2, %2, %3\n v_max_f64 %3, %3, %0\n")
KERNEL_D(k_rnd64,  "v_rndne_f64 %0, %0\n v_floor_f64 %1, %1\n v_trunc_f64 %2, %2\n v_fract_f64 %3, %3\n")
KERNEL_D(k_dpp,    "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 row_ror:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_bcast:15 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %4 row_bcast:31 row_mask:0xf bank_mask:0xf\n")
KERNEL_D(k_bperm,  "ds_bpermute_b32 %4, %5, %4\n ds_bpermute_b32 %5, %6, %5\n ds_bpermute_b32 %6, %7, %6\n ds_bpermute_b32 %7, %4, %7\n s_waitcnt lgkmcnt(0)\n")
KERNEL_D(k_divfix, "v_div_fixup_f64 %0, %0, %1, %2\n v_div_fmas_f64 %1, %1, %2, %3\n v_div_fixup_f64 %2, %2, %3, %0\n v_div_fmas_f64 %3, %3, %0, %1\n")

template <typename K>
double run(K k, double* out, int bpc = 8) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2048;
    k<<<256 * bpc, 256>>>(out, 16); hipDeviceSynchronize();
    hipEventRecord(a); k<<<256 * bpc, 256>>>(out, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    // instructions per SIMD: 8 waves x iters x 16 ; cycles at 2.4 GHz
    return ms * 1e-3 * 2.4e9 / ((double)bpc * iters * 16);
}

int main() {
    double* out; hipMalloc(&out, 8ull * 256 * 8 * 256);
    const double base = run(k_fma64, out);
    printf("v_fma_f64: %.2f cycles/instr at 2.4 GHz (= 4.00 if the clock holds)\n", base);
for (int bpc : {1, 2, 4, 8}) printf("waves/SIMD %d: fma64 4 chains %.2f  2 chains %.2f  1 chain %.2f cycles/instr/SIMD (2.4 GHz units)\n", bpc, run(k_fma64, out, bpc), run(k_fma64_2chain, out, bpc), run(k_fma64_1chain, out, bpc));
#define SHOW(k) printf("%-10s %.2f x fma\n", #k, run(k, out) / base);
    SHOW(k_fma64) SHOW(k_add64) SHOW(k_mul64) SHOW(k_ldexp) SHOW(k_rcp64) SHOW(k_rsq64) SHOW(k_sqrt64) SHOW(k_frexpm) SHOW(k_frexpe)
    SHOW(k_cvtdi) SHOW(k_cvtid) SHOW(k_cvtfd) SHOW(k_cvtdf) SHOW(k_rcp32) SHOW(k_log32) SHOW(k_exp32) SHOW(k_fma32) SHOW(k_mullo)
    SHOW(k_mulhi) SHOW(k_mad64) SHOW(k_mul24) SHOW(k_lshl64) SHOW(k_and32) SHOW(k_mov64) SHOW(k_cmp64) SHOW(k_min64) SHOW(k_rnd64)
    SHOW(k_dpp) SHOW(k_bperm) SHOW(k_divfix)
    return 0;
}
