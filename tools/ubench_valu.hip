// Microbenchmark: cycles per wave-instruction of the vector instructions the extractor's kernels are
// made of (the packed 16-bit min/max and add of the FAST compass test and arc score, byte permutes,
// byte alignment, 24-bit multiply-add, dot products of the resize and blur, popcount), register operands
// only, eight independent chains per wave, at 1, 2, 4 and 8 waves per SIMD on every compute unit.
// Prints clocks per wave-instruction per SIMD (from s_memtime inside the kernel, so independent of the
// clock the chip holds) and the chip-wide rate in T wave-instructions/s from the event time.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#define OPS_PER_ITER 64  // 8 chains x 8 rounds

// One round = the instruction once on each of the 8 chains r0..r7 (independent of each other).
#define ROUND(INS)                                                                                              \
    asm volatile(INS("%0") "\n" INS("%1") "\n" INS("%2") "\n" INS("%3") "\n" INS("%4") "\n" INS("%5") "\n" INS("%6") \
                     "\n" INS("%7")                                                                             \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)               \
                 : "v"(x), "v"(y), "s"(sc));

#define I_FMA(R) "v_fma_f32 " R ", " R ", %8, %9"
#define I_ADD(R) "v_add_u32 " R ", " R ", %8"
#define I_XOR(R) "v_xor_b32 " R ", " R ", %8"
#define I_BCNT(R) "v_bcnt_u32_b32 " R ", %8, " R
#define I_PKMAX(R) "v_pk_max_u16 " R ", " R ", %8"
#define I_PKMIN(R) "v_pk_min_u16 " R ", " R ", %8"
#define I_PKADD(R) "v_pk_add_u16 " R ", " R ", %8"
#define I_PKSUB(R) "v_pk_sub_i16 " R ", " R ", %8"
#define I_PERM(R) "v_perm_b32 " R ", " R ", %8, %9"
#define I_ALIGN(R) "v_alignbyte_b32 " R ", " R ", %8, 3"
#define I_MAD24(R) "v_mad_u32_u24 " R ", " R ", %8, %9"
#define I_MADI24(R) "v_mad_i32_i24 " R ", " R ", %8, %9"
#define I_DOT4(R) "v_dot4_u32_u8 " R ", " R ", %8, %9"
#define I_DOT2(R) "v_dot2_u32_u16 " R ", " R ", %8, %9"
#define I_MAX3(R) "v_max3_u32 " R ", " R ", %8, %9"
#define I_MED3(R) "v_med3_i32 " R ", " R ", %8, %9"
#define I_MINU(R) "v_min_u32 " R ", " R ", %8"
#define I_AND_OR(R) "v_and_or_b32 " R ", " R ", %8, %9"
#define I_LSHL_OR(R) "v_lshl_or_b32 " R ", " R ", 3, %8"
#define I_CNDMASK(R) "v_cndmask_b32 " R ", " R ", %8, vcc"
#define I_XOR_S(R) "v_xor_b32 " R ", %10, " R
#define I_PKFMA(R) "v_pk_mul_lo_u16 " R ", " R ", %8"
#define I_SAD(R) "v_sad_u8 " R ", " R ", %8, %9"
#define I_MQSAD(R) "v_msad_u8 " R ", " R ", %8, %9"
#define I_BFE(R) "v_bfe_u32 " R ", " R ", 3, 8"
#define I_MUL_LO(R) "v_mul_lo_u32 " R ", " R ", %8"
#define I_MUL24(R) "v_mul_u32_u24 " R ", " R ", %8"

#define I_AND(R) "v_and_b32 " R ", " R ", %8"
#define I_OR(R) "v_or_b32 " R ", " R ", %8"
#define I_SUB(R) "v_sub_u32 " R ", " R ", %8"
#define I_LSHL(R) "v_lshlrev_b32 " R ", 3, " R
#define I_LSHR(R) "v_lshrrev_b32 " R ", 3, " R
#define I_MAXF(R) "v_max_f32 " R ", " R ", %8"
#define I_MINF(R) "v_min_f32 " R ", " R ", %8"
#define I_MULF(R) "v_mul_f32 " R ", " R ", %8"
#define I_ADDF(R) "v_add_f32 " R ", " R ", %8"
#define I_MAXU(R) "v_max_u32 " R ", " R ", %8"
#define I_MAXI(R) "v_max_i32 " R ", " R ", %8"
#define I_PKFMAF32(R) "v_pk_fma_f32 %0, %0, %8, %9"
#define I_PKMAXF16(R) "v_pk_max_f16 " R ", " R ", %8"
#define I_PKMINF16(R) "v_pk_min_f16 " R ", " R ", %8"
#define I_PKADDF16(R) "v_pk_add_f16 " R ", " R ", %8"
#define I_PKFMAF16(R) "v_pk_fma_f16 " R ", " R ", %8, %9"
#define I_PKMAXI16(R) "v_pk_max_i16 " R ", " R ", %8"
#define I_MAXF16(R) "v_max_f16 " R ", " R ", %8"
#define I_MAXU16(R) "v_max_u16 " R ", " R ", %8"
#define I_ADDU16(R) "v_add_u16 " R ", " R ", %8"
#define I_CVTUB0(R) "v_cvt_f32_ubyte0 " R ", " R
#define I_MOV(R) "v_mov_b32 " R ", %8"
#define I_MAX3F(R) "v_max3_f32 " R ", " R ", %8, %9"
#define I_MIN3F(R) "v_min3_f32 " R ", " R ", %8, %9"
#define I_MED3F(R) "v_med3_f32 " R ", " R ", %8, %9"
#define I_ADD3(R) "v_add3_u32 " R ", " R ", %8, %9"
#define I_XOR3(R) "v_xor3_b32 " R ", " R ", %8, %9"
#define I_OR3(R) "v_or3_b32 " R ", " R ", %8, %9"
#define I_LSHLADD(R) "v_lshl_add_u32 " R ", " R ", 3, %8"
#define I_AND_LIT(R) "v_and_b32 " R ", 0x7f7f7f7f, " R
#define I_AND_INL(R) "v_and_b32 " R ", 15, " R
#define I_ADD_S(R) "v_add_u32 " R ", %10, " R
#define I_SUB_LIT(R) "v_sub_u32 " R ", 0x80808080, " R
#define I_CNDMASK_S(R) "v_cndmask_b32 " R ", " R ", %8, s[10:11]"
#define I_CMP(R) "v_cmp_lt_u32 vcc, " R ", %8"
#define I_PKLSHL(R) "v_pk_lshlrev_b16 " R ", 1, " R
#define I_SUBREV(R) "v_subrev_u32 " R ", " R ", %8"
#define I_BFI(R) "v_bfi_b32 " R ", " R ", %8, %9"
#define I_ALIGNBIT(R) "v_alignbit_b32 " R ", " R ", %8, 8"
#define I_SDWA_MAX(R) "v_max_u16_sdwa " R ", " R ", %8 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"
#define I_ADD_SDWA(R) "v_add_u32_sdwa " R ", " R ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"

#define KERNEL(NAME, INS)                                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, long long *clk, uint32_t seed, int iters) \
    {                                                                                                  \
        uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, \
                 r6 = r0 * 17u, r7 = r0 * 19u;                                                         \
        const uint32_t x = seed * 0x9E3779B9u + threadIdx.x, y = 0x07050301u ^ (seed & 0x03030303u);   \
        const uint32_t sc = seed * 77u;                                                                \
        const long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int it = 0; it < iters; it++) {                                                           \
            ROUND(INS) ROUND(INS) ROUND(INS) ROUND(INS) ROUND(INS) ROUND(INS) ROUND(INS) ROUND(INS)    \
        }                                                                                              \
        const long long t1 = __builtin_amdgcn_s_memtime();                                             \
        out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                   \
        if ((threadIdx.x & 63) == 0)                                                                   \
            clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                        \
    }

KERNEL(fma_f32, I_FMA)
KERNEL(add_u32, I_ADD)
KERNEL(xor_b32, I_XOR)
KERNEL(xor_b32_sgpr, I_XOR_S)
KERNEL(bcnt_u32_b32, I_BCNT)
KERNEL(pk_max_u16, I_PKMAX)
KERNEL(pk_min_u16, I_PKMIN)
KERNEL(pk_add_u16, I_PKADD)
KERNEL(pk_sub_i16, I_PKSUB)
KERNEL(pk_mul_lo_u16, I_PKFMA)
KERNEL(perm_b32, I_PERM)
KERNEL(alignbyte_b32, I_ALIGN)
KERNEL(mad_u32_u24, I_MAD24)
KERNEL(mad_i32_i24, I_MADI24)
KERNEL(mul_u32_u24, I_MUL24)
KERNEL(mul_lo_u32, I_MUL_LO)
KERNEL(dot4_u32_u8, I_DOT4)
KERNEL(dot2_u32_u16, I_DOT2)
KERNEL(max3_u32, I_MAX3)
KERNEL(med3_i32, I_MED3)
KERNEL(min_u32, I_MINU)
KERNEL(and_or_b32, I_AND_OR)
KERNEL(lshl_or_b32, I_LSHL_OR)
KERNEL(cndmask_b32, I_CNDMASK)
KERNEL(sad_u8, I_SAD)
KERNEL(msad_u8, I_MQSAD)
KERNEL(bfe_u32, I_BFE)

KERNEL(and_b32, I_AND)
KERNEL(or_b32, I_OR)
KERNEL(sub_u32, I_SUB)
KERNEL(subrev_u32, I_SUBREV)
KERNEL(lshlrev_b32, I_LSHL)
KERNEL(lshrrev_b32, I_LSHR)
KERNEL(max_f32, I_MAXF)
KERNEL(min_f32, I_MINF)
KERNEL(mul_f32, I_MULF)
KERNEL(add_f32, I_ADDF)
KERNEL(max_u32, I_MAXU)
KERNEL(max_i32, I_MAXI)
KERNEL(pk_max_f16, I_PKMAXF16)
KERNEL(pk_min_f16, I_PKMINF16)
KERNEL(pk_add_f16, I_PKADDF16)
KERNEL(pk_fma_f16, I_PKFMAF16)
KERNEL(pk_max_i16, I_PKMAXI16)
KERNEL(max_f16, I_MAXF16)
KERNEL(max_u16, I_MAXU16)
KERNEL(add_u16, I_ADDU16)
KERNEL(cvt_f32_ubyte0, I_CVTUB0)
KERNEL(mov_b32, I_MOV)
KERNEL(max3_f32, I_MAX3F)
KERNEL(min3_f32, I_MIN3F)
KERNEL(med3_f32, I_MED3F)
KERNEL(add3_u32, I_ADD3)
KERNEL(or3_b32, I_OR3)
KERNEL(lshl_add_u32, I_LSHLADD)
KERNEL(cndmask_sgpr, I_CNDMASK_S)
KERNEL(and_b32_literal, I_AND_LIT)
KERNEL(and_b32_inline, I_AND_INL)
KERNEL(add_u32_sgpr, I_ADD_S)
KERNEL(sub_u32_literal, I_SUB_LIT)
KERNEL(cmp_lt_u32, I_CMP)
KERNEL(pk_lshlrev_b16, I_PKLSHL)
KERNEL(bfi_b32, I_BFI)
KERNEL(alignbit_b32, I_ALIGNBIT)
KERNEL(max_u16_sdwa, I_SDWA_MAX)
KERNEL(add_u32_sdwa, I_ADD_SDWA)

// mixes of the two issue classes: whole rounds (eight independent chains) of one instruction, then of another
#define KERNEL_MIX(NAME, SEQ)                                                                          \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, long long *clk, uint32_t seed, int iters) \
    {                                                                                                  \
        uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, \
                 r6 = r0 * 17u, r7 = r0 * 19u;                                                         \
        const uint32_t x = seed * 0x9E3779B9u + threadIdx.x, y = 0x07050301u ^ (seed & 0x03030303u);   \
        const uint32_t sc = seed * 77u;                                                                \
        const long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int it = 0; it < iters; it++) {                                                           \
            SEQ                                                                                        \
        }                                                                                              \
        const long long t1 = __builtin_amdgcn_s_memtime();                                             \
        out[blockIdx.x * 256 + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                   \
        if ((threadIdx.x & 63) == 0)                                                                   \
            clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                        \
    }
KERNEL_MIX(mix_1add_1pkmax, ROUND(I_ADD) ROUND(I_PKMAX) ROUND(I_ADD) ROUND(I_PKMAX) ROUND(I_ADD) ROUND(I_PKMAX) ROUND(I_ADD) ROUND(I_PKMAX))
KERNEL_MIX(mix_3add_1pkmax, ROUND(I_ADD) ROUND(I_XOR) ROUND(I_SUB) ROUND(I_PKMAX) ROUND(I_ADD) ROUND(I_XOR) ROUND(I_SUB) ROUND(I_PKMAX))
KERNEL_MIX(mix_1and_1perm, ROUND(I_AND) ROUND(I_PERM) ROUND(I_AND) ROUND(I_PERM) ROUND(I_AND) ROUND(I_PERM) ROUND(I_AND) ROUND(I_PERM))


// Scalar instructions: eight independent scalar chains per wave (one scalar unit serves the four SIMDs of a
// compute unit), alone and beside vector instructions of the same wave.
#define SROUND(INS)                                                                                             \
    asm volatile(INS("%0") "\n" INS("%1") "\n" INS("%2") "\n" INS("%3") "\n" INS("%4") "\n" INS("%5") "\n" INS("%6") \
                     "\n" INS("%7")                                                                             \
                 : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3), "+s"(q4), "+s"(q5), "+s"(q6), "+s"(q7)               \
                 : "s"(sc)                                                                                     \
                 : "s20", "s21", "scc");
#define S_ADD(R) "s_add_u32 " R ", " R ", %8"
#define S_AND(R) "s_and_b32 " R ", " R ", %8"
#define S_MUL(R) "s_mul_i32 " R ", " R ", %8"
#define S_WAIT(R) "s_waitcnt lgkmcnt(0)"
#define S_NOP(R) "s_nop 0"
#define KERNEL_SMIX(NAME, SEQ)                                                                         \
    __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, long long *clk, uint32_t seed, int iters) \
    {                                                                                                  \
        uint32_t r0 = seed + threadIdx.x, r1 = r0 * 3u, r2 = r0 * 5u, r3 = r0 * 7u, r4 = r0 * 11u, r5 = r0 * 13u, \
                 r6 = r0 * 17u, r7 = r0 * 19u;                                                         \
        uint32_t q0 = seed, q1 = seed * 3u, q2 = seed * 5u, q3 = seed * 7u, q4 = seed * 11u, q5 = seed * 13u,  \
                 q6 = seed * 17u, q7 = seed * 19u;                                                     \
        const uint32_t x = seed * 0x9E3779B9u + threadIdx.x, y = 0x07050301u ^ (seed & 0x03030303u);   \
        const uint32_t sc = seed * 77u;                                                                \
        const long long t0 = __builtin_amdgcn_s_memtime();                                             \
        for (int it = 0; it < iters; it++) {                                                           \
            SEQ                                                                                        \
        }                                                                                              \
        const long long t1 = __builtin_amdgcn_s_memtime();                                             \
        out[blockIdx.x * 256 + threadIdx.x] =                                                          \
            r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ q0 ^ q1 ^ q2 ^ q3 ^ q4 ^ q5 ^ q6 ^ q7;              \
        if ((threadIdx.x & 63) == 0)                                                                   \
            clk[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                        \
    }
#define S_MOV(R) "s_mov_b32 " R ", 0x12345678"
#define S_SAVEEXEC(R) "s_and_saveexec_b64 s[20:21], vcc\n s_or_b64 exec, exec, s[20:21]"
#define S_CMPSEL(R) "s_cmp_lt_u32 " R ", %8\n s_cselect_b32 " R ", " R ", %8"
KERNEL_SMIX(s_mov_b32, SROUND(S_MOV) SROUND(S_MOV) SROUND(S_MOV) SROUND(S_MOV) SROUND(S_MOV) SROUND(S_MOV) SROUND(S_MOV) SROUND(S_MOV))
// two instructions per expansion: the rate printed counts expansions
KERNEL_SMIX(s_saveexec_or, SROUND(S_SAVEEXEC) SROUND(S_SAVEEXEC) SROUND(S_SAVEEXEC) SROUND(S_SAVEEXEC) SROUND(S_SAVEEXEC) SROUND(S_SAVEEXEC) SROUND(S_SAVEEXEC) SROUND(S_SAVEEXEC))
KERNEL_SMIX(s_cmp_cselect, SROUND(S_CMPSEL) SROUND(S_CMPSEL) SROUND(S_CMPSEL) SROUND(S_CMPSEL) SROUND(S_CMPSEL) SROUND(S_CMPSEL) SROUND(S_CMPSEL) SROUND(S_CMPSEL))
KERNEL_SMIX(s_add_u32, SROUND(S_ADD) SROUND(S_ADD) SROUND(S_ADD) SROUND(S_ADD) SROUND(S_ADD) SROUND(S_ADD) SROUND(S_ADD) SROUND(S_ADD))
KERNEL_SMIX(s_mul_i32, SROUND(S_MUL) SROUND(S_MUL) SROUND(S_MUL) SROUND(S_MUL) SROUND(S_MUL) SROUND(S_MUL) SROUND(S_MUL) SROUND(S_MUL))
KERNEL_SMIX(s_waitcnt, SROUND(S_WAIT) SROUND(S_WAIT) SROUND(S_WAIT) SROUND(S_WAIT) SROUND(S_WAIT) SROUND(S_WAIT) SROUND(S_WAIT) SROUND(S_WAIT))
KERNEL_SMIX(s_nop, SROUND(S_NOP) SROUND(S_NOP) SROUND(S_NOP) SROUND(S_NOP) SROUND(S_NOP) SROUND(S_NOP) SROUND(S_NOP) SROUND(S_NOP))
// per_op 1: the rate printed counts the vector instructions only (32 per iteration beside 32 scalar ones)
KERNEL_SMIX(mix_1sadd_1pkmax, ROUND(I_PKMAX) SROUND(S_ADD) ROUND(I_PKMAX) SROUND(S_ADD) ROUND(I_PKMAX) SROUND(S_ADD) ROUND(I_PKMAX) SROUND(S_ADD))
KERNEL_SMIX(mix_1sadd_1add, ROUND(I_ADD) SROUND(S_ADD) ROUND(I_ADD) SROUND(S_ADD) ROUND(I_ADD) SROUND(S_ADD) ROUND(I_ADD) SROUND(S_ADD))

typedef void (*kern_t)(uint32_t *, long long *, uint32_t, int);
struct Entry {
    const char *name;
    kern_t fn;
    int per_op;  // instructions per macro expansion
};
#define E(NAME) {#NAME, k_##NAME, 1}
static const Entry entries[] = {E(fma_f32),      E(add_u32),      E(xor_b32),      E(xor_b32_sgpr), E(bcnt_u32_b32),
                                E(pk_max_u16),   E(pk_min_u16),   E(pk_add_u16),   E(pk_sub_i16),   E(pk_mul_lo_u16),
                                E(perm_b32),     E(alignbyte_b32), E(mad_u32_u24), E(mad_i32_i24),  E(mul_u32_u24),
                                E(mul_lo_u32),   E(dot4_u32_u8),  E(dot2_u32_u16), E(max3_u32),     E(med3_i32),
                                E(min_u32),      E(and_or_b32),   E(lshl_or_b32),  E(cndmask_b32),  E(sad_u8),
                                E(msad_u8),      E(bfe_u32),
    E(and_b32), E(or_b32), E(sub_u32), E(subrev_u32), E(lshlrev_b32), E(lshrrev_b32), E(max_f32), E(min_f32), E(mul_f32), E(add_f32), E(max_u32), E(max_i32), E(pk_max_f16), E(pk_min_f16), E(pk_add_f16), E(pk_fma_f16), E(pk_max_i16), E(max_f16), E(max_u16), E(add_u16), E(cvt_f32_ubyte0), E(mov_b32), E(max3_f32), E(min3_f32), E(med3_f32), E(add3_u32), E(or3_b32), E(lshl_add_u32), E(cndmask_sgpr), E(and_b32_literal), E(and_b32_inline), E(add_u32_sgpr), E(sub_u32_literal), E(cmp_lt_u32), E(pk_lshlrev_b16), E(bfi_b32), E(alignbit_b32), E(max_u16_sdwa), E(add_u32_sdwa),
    E(mix_1add_1pkmax), E(mix_3add_1pkmax), E(mix_1and_1perm),
    E(s_add_u32), E(s_mov_b32), E(s_saveexec_or), E(s_cmp_cselect), E(s_mul_i32), E(s_waitcnt), E(s_nop), E(mix_1sadd_1pkmax), E(mix_1sadd_1add)};

int main(int argc, char **argv)
{
    const int ncu = 256, iters = 2048;
    const int max_blocks = ncu * 8;
    uint32_t *d_out;
    long long *d_clk;
    hipMalloc(&d_out, (size_t)max_blocks * 256 * 4);
    hipMalloc(&d_clk, (size_t)max_blocks * 4 * 8);
    std::vector<long long> clk((size_t)max_blocks * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("%-16s %s\n", "instruction", "waves/SIMD: clocks per wave-instruction per SIMD (in-kernel) | T wave-instr/s (events)");
    for (const Entry &en : entries) {
        if (argc > 1 && strcmp(argv[1], en.name) != 0)
            continue;
        printf("%-16s", en.name);
        for (int wps = 1; wps <= 8; wps *= 2) {
            // a 256-thread workgroup puts one wave on each SIMD: wps workgroups per compute unit
            const int blocks = ncu * wps;
            float best_ms = 1e30f;
            double cyc = 0;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                en.fn<<<blocks, 256>>>(d_out, d_clk, 12345u + rep, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best_ms) {
                    best_ms = ms;
                    hipMemcpy(clk.data(), d_clk, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
                    double s = 0;
                    for (int i = 0; i < blocks * 4; i++)
                        s += (double)clk[i];
                    // a SIMD holds wps waves; while they run together it issues wps * ops instructions
                    // in the time one of them takes
                    cyc = s / (blocks * 4) / ((double)iters * OPS_PER_ITER * en.per_op) / wps;
                }
            }
            const double winstr = (double)blocks * 4 * iters * OPS_PER_ITER * en.per_op;
            printf("  %d: %5.2f | %6.3f", wps, cyc, winstr / (best_ms * 1e-3) / 1e12);
        }
        printf("\n");
    }
    return 0;
}
