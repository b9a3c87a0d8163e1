// recompute_fstat.cuh — the feature-stationary fused kernel of the recompute search (included by recompute.hip only).
// fused_fstat_kernel<KS, TILED>: encode GEMM + row norms + feature-space scoring + candidate emission in one persistent
// kernel whose MFMA loops are hand-scheduled inline asm; tile_features_kernel: the fragment-major feature
// copy it reads; fold_candidates_kernel: merges the emitted candidates into the running best-k.
// Reference arithmetic replaced: RecomputeSearcher::search src/index/recompute.rs:86-109 with the provider tail of
// src/embedding/candle.rs:165,218-225.  scripts/check_fstat_asm.py guards the hand-counted LDS reads (CPU test).
#pragma once
#include "common.cuh"
#include "internal.h"
#include <utility>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// compile-time loop: the body sees its index as an integral_constant (inline-asm "i" operands need true constants)
template <class Fn, int... I>
__device__ __forceinline__ void static_for_impl(Fn &&fn, std::integer_sequence<int, I...>) { (fn(std::integral_constant<int, I>{}), ...); }
template <int N, class Fn>
__device__ __forceinline__ void static_for(Fn &&fn) { static_for_impl(fn, std::make_integer_sequence<int, N>{}); }


// ------------------------------------------------------------------------------------------------
// fused_fstat_kernel<KS>: the fused search pass with the FEATURES stationary in registers (h = 16*KS <= 256, L = 1).
// encode_kernel<.., true, false> re-reads every operand fragment from LDS for each MFMA group (88 KB of LDS reads per
// k-step per CU) and re-streams all of W (+G) from L2 for every 128 passages.  Here:
//   * persistent workgroup of 4 waves, ONE wave per SIMD (512 registers each); unit = 4 x RB x 32 passages;
//   * a wave keeps the feature fragments of its RB x 32 passages for the whole K in registers (RB x KS x 4 AGPRs, the B operand:
//     every tile is computed as D[column or query][passage]), so one 16-B LDS read (a weight / G fragment, the A operand) feeds RB MFMAs;
//   * W / G stream through a double-buffered LDS image in sub-slices of 128 columns (KS x 4 KiB, LDS-DMA with scalar
//     base + per-lane offset, swizzled on the source side), one barrier per sub-slice;
//   * per unit: dp/128 W sub-slices (4 column tiles each: MFMAs into ping-pong accumulators; the squares of a finished
//     tile are folded into per-lane sums of squares in the shadow of the next tile's MFMAs) -> row norms (a lane's registers are
//     columns of one passage: in-lane sum + one cross-half exchange, once per unit) -> nqt G sub-slices (query tile x {hi, lo, lo2} accumulated into one score tile; A = G fragment, B = the
//     resident feature fragment) -> S[q][passage] = acc * (1 / norm); the next unit's features are fetched during the
//     last G piece, each register right after its last use;
//   * the weight fragments run through a 4-deep register ring (explicit software pipeline: with one wave per SIMD nobody
//     else hides the LDS latency); the loops are inline asm, one statement per k-step (see the FSTAT_* macros);
//   * weight visits, score visits and the unit's last score visit (which refills the features) are three instantiations of one
//     visit lambda called in sequence, so that no loop-carried register shuffle surrounds the refill.
// ------------------------------------------------------------------------------------------------
#ifdef LEANN_STAMPS // diagnostic build (scripts/stamps.sh): cycles per phase summed over wave 0 of every workgroup
__device__ unsigned long long g_fstat_stamps[16];
extern "C" int leann_debug_fstat_stamps(unsigned long long *out16, int reset) {
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fstat_stamps), 128) != hipSuccess) return 1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_fstat_stamps), z, 128) != hipSuccess) return 1; }
    return 0;
}
#endif
// hand-counted LDS fragment reads (see the W loop): form (ii) of the guide's inline-asm rules — "=v" load, then a wait
// statement that names the destination "+v" before its first consumer
#define FSTAT_DS_READ(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off) : "memory")
#define FSTAT_LGKM_WAIT(cnt, dst) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "i"(cnt) : "memory")
// One asm statement per k-step: counted LDS wait, the ring's next fragment read, RB (= 2) MFMAs.  Explicit register classes: the
// resident feature fragments live in AGPRs ("a"), the accumulators in arch VGPRs ("v") where the VALU epilogues (squares,
// scaling) read them without v_accvgpr_read copies.  No s_nop between MFMAs (a pad inside an MFMA-paced stream costs 17-43
// cycles): hipcc does not know these are MFMAs, so (a) every VALU reader of an accumulator is kept >= 4 MFMAs behind its last
// write and (b) the feature fragments are pinned into their AGPRs at the top of a unit, far from their first MFMA.
#define FSTAT_MM "v_mfma_f32_32x32x16_bf16 "
// W: acc += A(feature frag, AGPR) x B(weight frag, VGPR).  Variants: Z = accumulators start from 0; RD / NR = with / without the
// ring's next read; SQ1 / SQ3 = one / three squares of the PREVIOUS tile behind each MFMA as scalar v_fma_f32 fillers (free in an
// MFMA gap when hand-placed; the compiler's own placement packs them into v_pk_fma_f32, which costs ~19 cycles per MFMA here).
#define FSTAT_W_HEAD_RD "s_waitcnt lgkmcnt(%[cnt])\n\tds_read_b128 %[nb], %[addr] offset:%[off]\n\t"
#define FSTAT_W_HEAD_NR "s_waitcnt lgkmcnt(%[cnt])\n\t"
#define FSTAT_SQ(q, p) "v_fma_f32 %[" #q "], %[" #p "], %[" #p "], %[" #q "]\n\t"
#define FSTAT_W_RDZ(C0_, C1_, A0_, A1_, B_, NB_, ADDR_, OFF_, CNT_)                                                                    \
    asm volatile(FSTAT_W_HEAD_RD FSTAT_MM "%[x0], %[bb], %[f0], 0\n\t" FSTAT_MM "%[x1], %[bb], %[f1], 0"                       \
                 : [x0] "=&v"(C0_), [x1] "=&v"(C1_), [bb] "+v"(B_), [nb] "=&v"(NB_)                                                \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [addr] "v"(ADDR_), [off] "i"(OFF_), [cnt] "i"(CNT_) : "memory")
#define FSTAT_W_RD(C0_, C1_, A0_, A1_, B_, NB_, ADDR_, OFF_, CNT_)                                                                     \
    asm volatile(FSTAT_W_HEAD_RD FSTAT_MM "%[x0], %[bb], %[f0], %[x0]\n\t" FSTAT_MM "%[x1], %[bb], %[f1], %[x1]"               \
                 : [x0] "+v"(C0_), [x1] "+v"(C1_), [bb] "+v"(B_), [nb] "=&v"(NB_)                                                  \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [addr] "v"(ADDR_), [off] "i"(OFF_), [cnt] "i"(CNT_) : "memory")
#define FSTAT_W_RD_SQ1(C0_, C1_, A0_, A1_, B_, NB_, ADDR_, OFF_, CNT_, Q0_, P0_, Q1_, P1_)                                                 \
    asm volatile(FSTAT_W_HEAD_RD FSTAT_MM "%[x0], %[bb], %[f0], %[x0]\n\t" FSTAT_SQ(q0, p0) FSTAT_MM "%[x1], %[bb], %[f1], %[x1]\n\t" FSTAT_SQ(q1, p1) \
                 : [x0] "+v"(C0_), [x1] "+v"(C1_), [bb] "+v"(B_), [nb] "=&v"(NB_), [q0] "+v"(Q0_), [q1] "+v"(Q1_)                     \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [addr] "v"(ADDR_), [off] "i"(OFF_), [cnt] "i"(CNT_), [p0] "v"(P0_), [p1] "v"(P1_) : "memory")
#define FSTAT_W_NR_SQ1(C0_, C1_, A0_, A1_, B_, CNT_, Q0_, P0_, Q1_, P1_)                                                                \
    asm volatile(FSTAT_W_HEAD_NR FSTAT_MM "%[x0], %[bb], %[f0], %[x0]\n\t" FSTAT_SQ(q0, p0) FSTAT_MM "%[x1], %[bb], %[f1], %[x1]\n\t" FSTAT_SQ(q1, p1) \
                 : [x0] "+v"(C0_), [x1] "+v"(C1_), [bb] "+v"(B_), [q0] "+v"(Q0_), [q1] "+v"(Q1_)                                     \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [cnt] "i"(CNT_), [p0] "v"(P0_), [p1] "v"(P1_) : "memory")
#define FSTAT_W_SQ3_BODY FSTAT_MM "%[x0], %[bb], %[f0], %[x0]\n\t" FSTAT_SQ(q0, p0) FSTAT_SQ(q2, p2) FSTAT_SQ(q4, p4)         \
                         FSTAT_MM "%[x1], %[bb], %[f1], %[x1]\n\t" FSTAT_SQ(q1, p1) FSTAT_SQ(q3, p3) FSTAT_SQ(q5, p5)
#define FSTAT_W_RD_SQ3(C0_, C1_, A0_, A1_, B_, NB_, ADDR_, OFF_, CNT_, Q0_, P0_, Q1_, P1_, Q2_, P2_, Q3_, P3_, Q4_, P4_, Q5_, P5_)                 \
    asm volatile(FSTAT_W_HEAD_RD FSTAT_W_SQ3_BODY                                                                            \
                 : [x0] "+v"(C0_), [x1] "+v"(C1_), [bb] "+v"(B_), [nb] "=&v"(NB_), [q0] "+v"(Q0_), [q1] "+v"(Q1_), [q2] "+v"(Q2_),     \
                   [q3] "+v"(Q3_), [q4] "+v"(Q4_), [q5] "+v"(Q5_)                                                                \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [addr] "v"(ADDR_), [off] "i"(OFF_), [cnt] "i"(CNT_), [p0] "v"(P0_), [p1] "v"(P1_),  \
                   [p2] "v"(P2_), [p3] "v"(P3_), [p4] "v"(P4_), [p5] "v"(P5_) : "memory")
#define FSTAT_W_NR_SQ3(C0_, C1_, A0_, A1_, B_, CNT_, Q0_, P0_, Q1_, P1_, Q2_, P2_, Q3_, P3_, Q4_, P4_, Q5_, P5_)                                \
    asm volatile(FSTAT_W_HEAD_NR FSTAT_W_SQ3_BODY                                                                            \
                 : [x0] "+v"(C0_), [x1] "+v"(C1_), [bb] "+v"(B_), [q0] "+v"(Q0_), [q1] "+v"(Q1_), [q2] "+v"(Q2_), [q3] "+v"(Q3_),      \
                   [q4] "+v"(Q4_), [q5] "+v"(Q5_)                                                                              \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [cnt] "i"(CNT_), [p0] "v"(P0_), [p1] "v"(P1_), [p2] "v"(P2_), [p3] "v"(P3_),        \
                   [p4] "v"(P4_), [p5] "v"(P5_) : "memory")
// ... D variants carry one 1-KiB LDS-DMA piece of the NEXT sub-slice between the two MFMAs (M0 = LDS destination, written in the
// same statement; source = scalar base + per-lane offset + immediate, the immediate also offsets the LDS side).  Spread one
// piece per four k-steps the DMA issue hides in the MFMA gap (+0.5 cycles per MFMA); issued back to back at the top of a visit
// the 64 pieces of a sub-slice cost ~1 000 cycles (the CU's 64 B/clk vector-memory path).
#define FSTAT_DMA "s_mov_b32 m0, %[ld]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vo], %[sb] offset:%[doff]\n\t"
#define FSTAT_W_RD_D(C0_, C1_, A0_, A1_, B_, NB_, ADDR_, OFF_, CNT_, VO_, SB_, LD_, DOFF_)                                    \
    asm volatile(FSTAT_W_HEAD_RD FSTAT_MM "%[x0], %[bb], %[f0], %[x0]\n\t" FSTAT_DMA FSTAT_MM "%[x1], %[bb], %[f1], %[x1]"    \
                 : [x0] "+v"(C0_), [x1] "+v"(C1_), [bb] "+v"(B_), [nb] "=&v"(NB_)                                              \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [addr] "v"(ADDR_), [off] "i"(OFF_), [cnt] "i"(CNT_), [vo] "v"(VO_), [sb] "s"(SB_), \
                   [ld] "s"(LD_), [doff] "i"(DOFF_) : "memory")
#define FSTAT_W_RD_SQ1_D(C0_, C1_, A0_, A1_, B_, NB_, ADDR_, OFF_, CNT_, Q0_, P0_, Q1_, P1_, VO_, SB_, LD_, DOFF_)            \
    asm volatile(FSTAT_W_HEAD_RD FSTAT_MM "%[x0], %[bb], %[f0], %[x0]\n\t" FSTAT_SQ(q0, p0) FSTAT_DMA                        \
                 FSTAT_MM "%[x1], %[bb], %[f1], %[x1]\n\t" FSTAT_SQ(q1, p1)                                                  \
                 : [x0] "+v"(C0_), [x1] "+v"(C1_), [bb] "+v"(B_), [nb] "=&v"(NB_), [q0] "+v"(Q0_), [q1] "+v"(Q1_)               \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [addr] "v"(ADDR_), [off] "i"(OFF_), [cnt] "i"(CNT_), [p0] "v"(P0_), [p1] "v"(P1_), \
                   [vo] "v"(VO_), [sb] "s"(SB_), [ld] "s"(LD_), [doff] "i"(DOFF_) : "memory")
#define FSTAT_W_NR_SQ1_D(C0_, C1_, A0_, A1_, B_, CNT_, Q0_, P0_, Q1_, P1_, VO_, SB_, LD_, DOFF_)                              \
    asm volatile(FSTAT_W_HEAD_NR FSTAT_MM "%[x0], %[bb], %[f0], %[x0]\n\t" FSTAT_SQ(q0, p0) FSTAT_DMA                        \
                 FSTAT_MM "%[x1], %[bb], %[f1], %[x1]\n\t" FSTAT_SQ(q1, p1)                                                  \
                 : [x0] "+v"(C0_), [x1] "+v"(C1_), [bb] "+v"(B_), [q0] "+v"(Q0_), [q1] "+v"(Q1_)                                \
                 : [f0] "a"(A0_), [f1] "a"(A1_), [cnt] "i"(CNT_), [p0] "v"(P0_), [p1] "v"(P1_), [vo] "v"(VO_), [sb] "s"(SB_),  \
                   [ld] "s"(LD_), [doff] "i"(DOFF_) : "memory")
// One statement per k-step of the score-tile loop: the three pieces (hi, lo, lo2) of the k-step, their ring reads, six MFMAs and one
// LDS-DMA piece.  S0/S1/S2 = ring slots of this k-step's fragments (S0 and S1 are re-filled in place with fragments f + 5, f + 6 once
// their MFMAs have issued), S4 = the free slot taking fragment f + 4.  Z: the accumulators start from 0; T1 / T0: the last two
// k-steps (two / no reads left).
#define FSTAT_GK_MM(slot, c) FSTAT_MM "%[x0], %[" #slot "], %[f0], " c "\n\t"
#define FSTAT_GK_OPS_OUT(C0_, C1_, S0_, S1_, S2_) [x0] "+v"(C0_), [x1] "+v"(C1_), [s0] "+v"(S0_), [s1] "+v"(S1_), [s2] "+v"(S2_)
#define FSTAT_GK_OPS_IN(A0_, A1_, ADDR_, VO_, SB_, LD_, DOFF_) \
    [f0] "a"(A0_), [f1] "a"(A1_), [addr] "v"(ADDR_), [vo] "v"(VO_), [sb] "s"(SB_), [ld] "s"(LD_), [doff] "i"(DOFF_)
#define FSTAT_GK_BODY(Z0, Z1, RD0, RD1, RD2, W0, W1, W2)                                                                       \
    "s_waitcnt lgkmcnt(" W0 ")\n\t" RD0 FSTAT_MM "%[x0], %[s0], %[f0], " Z0 "\n\t" FSTAT_MM "%[x1], %[s0], %[f1], " Z1 "\n\t"   \
    "s_waitcnt lgkmcnt(" W1 ")\n\t" RD1 FSTAT_MM "%[x0], %[s1], %[f0], %[x0]\n\t" FSTAT_DMA FSTAT_MM "%[x1], %[s1], %[f1], %[x1]\n\t" \
    "s_waitcnt lgkmcnt(" W2 ")\n\t" RD2 FSTAT_MM "%[x0], %[s2], %[f0], %[x0]\n\t" FSTAT_MM "%[x1], %[s2], %[f1], %[x1]"
#define FSTAT_GK_RD(dst, off) "ds_read_b128 %[" #dst "], %[addr] offset:%[" #off "]\n\t"
#define FSTAT_GK_FULL(Z0, Z1, C0_, C1_, S0_, S1_, S2_, S4_, A0_, A1_, ADDR_, O4_, O5_, O6_, VO_, SB_, LD_, DOFF_)              \
    asm volatile(FSTAT_GK_BODY(Z0, Z1, FSTAT_GK_RD(s4, o4), FSTAT_GK_RD(s0, o5), FSTAT_GK_RD(s1, o6), "3", "3", "3")           \
                 : FSTAT_GK_OPS_OUT(C0_, C1_, S0_, S1_, S2_), [s4] "=&v"(S4_)                                                  \
                 : FSTAT_GK_OPS_IN(A0_, A1_, ADDR_, VO_, SB_, LD_, DOFF_), [o4] "i"(O4_), [o5] "i"(O5_), [o6] "i"(O6_) : "memory")
#define FSTAT_GK_FULL_Z(C0_, C1_, S0_, S1_, S2_, S4_, A0_, A1_, ADDR_, O4_, O5_, O6_, VO_, SB_, LD_, DOFF_) /* accumulators start from 0 */ \
    asm volatile(FSTAT_GK_BODY("0", "0", FSTAT_GK_RD(s4, o4), FSTAT_GK_RD(s0, o5), FSTAT_GK_RD(s1, o6), "3", "3", "3")           \
                 : [x0] "=&v"(C0_), [x1] "=&v"(C1_), [s0] "+v"(S0_), [s1] "+v"(S1_), [s2] "+v"(S2_), [s4] "=&v"(S4_)           \
                 : FSTAT_GK_OPS_IN(A0_, A1_, ADDR_, VO_, SB_, LD_, DOFF_), [o4] "i"(O4_), [o5] "i"(O5_), [o6] "i"(O6_) : "memory")
#define FSTAT_GK_T1(C0_, C1_, S0_, S1_, S2_, S4_, A0_, A1_, ADDR_, O4_, O5_, VO_, SB_, LD_, DOFF_)                              \
    asm volatile(FSTAT_GK_BODY("%[x0]", "%[x1]", FSTAT_GK_RD(s4, o4), FSTAT_GK_RD(s0, o5), "", "3", "3", "3")                   \
                 : FSTAT_GK_OPS_OUT(C0_, C1_, S0_, S1_, S2_), [s4] "=&v"(S4_)                                                  \
                 : FSTAT_GK_OPS_IN(A0_, A1_, ADDR_, VO_, SB_, LD_, DOFF_), [o4] "i"(O4_), [o5] "i"(O5_) : "memory")
#define FSTAT_GK_T0(C0_, C1_, S0_, S1_, S2_, A0_, A1_, ADDR_, VO_, SB_, LD_, DOFF_)                                            \
    asm volatile(FSTAT_GK_BODY("%[x0]", "%[x1]", "", "", "", "2", "1", "0")                                                   \
                 : FSTAT_GK_OPS_OUT(C0_, C1_, S0_, S1_, S2_)                                                                   \
                 : FSTAT_GK_OPS_IN(A0_, A1_, ADDR_, VO_, SB_, LD_, DOFF_) : "memory")
// Candidate emission (chunks after the first): CandEmit, internal.h.
#define FSTAT_MAX_QUERIES 256 // queries per launch: the encode GEMM is shared, each 32-query tile adds one G sub-slice per unit
#define LEANN_FSTAT_RB 2 // 32-passage blocks per wave (the per-k-step asm statements are written for two)
#ifndef LEANN_FSTAT_PFD
#define LEANN_FSTAT_PFD 0 // k-steps between a feature fragment's last MFMA and its refill for the next unit (experiment knob)
#endif
#ifndef LEANN_FSTAT_DB
#define LEANN_FSTAT_DB 0 // experiment: a SECOND set of feature registers, refilled for the next unit inside the first weight visit
#endif
// Fragment-major copy of the features for fused_fstat_kernel: Ft[block of 32 rows][k-step][lane = lh * 32 + row][8 bf16], i.e.
// the 1 KiB a wave loads per (block, k-step) is contiguous (8 full lines per instruction instead of 32 quarter lines of a
// row-major read: the row-major form costs ~170 issue cycles per load next to the MFMAs).  Rows are padded to whole units.
__global__ void tile_features_kernel(const uint16_t *__restrict__ F, uint64_t n, uint32_t h, uint64_t n_pad, uint4 *__restrict__ Ft) {
    const uint32_t ksn = h / 16;
    const uint64_t total = n_pad * (h / 8);
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t lane = (uint32_t)(idx & 63);
        const uint64_t t = idx >> 6;
        const uint32_t ks = (uint32_t)(t % ksn);
        const uint64_t row = (t / ksn) * 32 + (lane & 31);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < n) v = *reinterpret_cast<const uint4 *>(F + row * h + ks * 16 + (lane >> 5) * 8);
        Ft[idx] = v;
    }
}

// LIST (row-major features only): passage i of the launch is row idx[i] of F — the early filter of recompute.rs:62-79 (only the
// passages that pass the filter are embedded), with the allowed positions compacted ascending by scan.hip:compact_allow.  Slab
// columns and emitted keys are list indices; the finalize step maps them back to positions.
template <int KS, bool TILED, bool LIST = false>
__global__ void __launch_bounds__(256) fused_fstat_kernel(const uint16_t *__restrict__ F, uint64_t n, const uint16_t *__restrict__ Wp,
                                                          uint32_t dp, const uint16_t *__restrict__ Gp, uint32_t nq,
                                                          float *__restrict__ S, uint32_t n_rows_s, CandEmit em,
                                                          const uint32_t *__restrict__ idx = nullptr) {
    static_assert(!(TILED && LIST), "a row list gathers from the row-major features");
    constexpr int NWV = 4, RB = LEANN_FSTAT_RB; // waves per workgroup, 32-passage blocks per wave
    constexpr int SUB = 128;             // columns per W sub-slice (4 MFMA tiles)
    constexpr int KSB = SUB * 32;        // bytes of one k-step of a sub-slice (4 KiB)
    constexpr int SUBB = KS * KSB;       // bytes per sub-slice buffer (64 KiB at KS = 16)
    constexpr int RING = 4;              // weight fragments in flight per wave
    constexpr uint32_t H = KS * 16;
    constexpr uint32_t UNIT = NWV * RB * 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *sbuf = smem;                                     // [2][SUBB]
    float *sNrm = reinterpret_cast<float *>(smem + 2 * SUBB);       // [NWV][RB * 32] sums of squares of a wave's passages
    float *sThr = sNrm + NWV * RB * 32;                             // [FSTAT_MAX_QUERIES] emission thresholds of the launch's queries
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // wave-uniform in an SGPR: DMA addresses stay scalar + lane offset
    const int l31 = lane & 31, lh = lane >> 5;
    const int nqt = (int)((nq + 31) / 32);                 // 32-query tiles of this launch (<= FSTAT_MAX_QUERIES / 32)
    const int nsw = (int)(dp / SUB), nsub = nsw + nqt;
    const uint64_t n_units = (n + UNIT - 1) / UNIT;

    // DMA addresses = scalar base (SGPRs) + a 32-bit per-lane offset re-materialised at every use (the opaque asm keeps the
    // compiler from hoisting per-lane 64-bit pointers out of the loops, spilling them, and draining the DMA queue with a
    // vmcnt(0) in front of every reload).  Slot p of a 1-KiB piece holds source slot p ^ ((p >> 4) & 1).
    const uint32_t goff = (uint32_t)(lane ^ ((lane >> 4) & 1)) * 16;
    // Wave w stages k-steps w, w + 4, ...: the 1-KiB pieces of a k-step differ only by the instruction's immediate offset,
    // which the hardware adds to BOTH the global and the LDS address, so M0 (the LDS base) is written once per k-step —
    // every M0 write waits for the previous LDS-DMA to have consumed it (~175 cycles per DMA when each one has its own).
    auto stage = [&](int j, int buf) {
        unsigned char *dst = sbuf + buf * SUBB;
#pragma unroll
        for (int t = 0; t < KS / NWV; t++) {
            const int ks = wave + NWV * t;
            uint32_t vo = goff;
            asm volatile("" : "+v"(vo));
            __attribute__((address_space(3))) void *ldst = (__attribute__((address_space(3))) void *)(dst + ks * KSB);
            if (j < nsw) {
                const char *src = reinterpret_cast<const char *>(Wp) + ((size_t)ks * dp + (size_t)j * SUB) * 32 + vo;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, ldst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, ldst, 16, 1024, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, ldst, 16, 2048, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, ldst, 16, 3072, 0);
            } else { // query tile j - nsw: pieces hi, lo, lo2 = column tiles 0..2 of the image
                const char *src = reinterpret_cast<const char *>(Gp) + ((size_t)ks * nqt * 96 + (size_t)(j - nsw) * 96) * 32 + vo;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, ldst, 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, ldst, 16, 1024, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, ldst, 16, 2048, 0);
            }
        }
    };
    const uint32_t fro = (uint32_t)((2 * l31 + lh) ^ (((2 * l31 + lh) >> 4) & 1)) * 16; // fragment of column l31 within a 32-column tile
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem;
    uint32_t lrow[RB]; // LIST: the rows of F behind this lane's passages of the unit whose features are loaded next
    auto set_lrow = [&](uint64_t prow0) {
        if constexpr (LIST) {
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                const uint64_t row = prow0 + rb * 32 + l31;
                lrow[rb] = idx[row < n ? row : n - 1];
            }
        }
    };
    auto load_features = [&](bf16x8 (&a)[RB][KS], uint64_t prow0, int ks) {
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            if constexpr (TILED) { // F = fragment-major copy: one contiguous KiB per (block of 32 rows, k-step), padded to whole units
                a[rb][ks] = reinterpret_cast<const bf16x8 *>(F)[((prow0 / 32 + rb) * KS + ks) * 64 + lane];
            } else if constexpr (LIST) {
                a[rb][ks] = *reinterpret_cast<const bf16x8 *>(F + (uint64_t)lrow[rb] * H + ks * 16 + lh * 8);
            } else { // row-major; branch-free: rows past the end re-read the last row (their results are never stored or emitted)
                const uint64_t row = prow0 + rb * 32 + l31;
                a[rb][ks] = *reinterpret_cast<const bf16x8 *>(F + (row < n ? row : n - 1) * H + ks * 16 + lh * 8);
            }
        }
    };

    auto load_feature1 = [&](bf16x8 (&a)[RB][KS], uint64_t prow0, int rb, int ks) __attribute__((always_inline)) {
        if constexpr (TILED) a[rb][ks] = reinterpret_cast<const bf16x8 *>(F)[((prow0 / 32 + rb) * KS + ks) * 64 + lane];
        else if constexpr (LIST) a[rb][ks] = *reinterpret_cast<const bf16x8 *>(F + (uint64_t)lrow[rb] * H + ks * 16 + lh * 8);
        else {
            const uint64_t row = prow0 + rb * 32 + l31;
            a[rb][ks] = *reinterpret_cast<const bf16x8 *>(F + (row < n ? row : n - 1) * H + ks * 16 + lh * 8);
        }
    };
    if (blockIdx.x >= n_units) return;
    if (em.thr) sThr[tid] = em.thr[tid]; // 256 threads, FSTAT_MAX_QUERIES = 256 slots; // visible to every wave long before its first use (a barrier per sub-slice)
    constexpr bool DB = LEANN_FSTAT_DB && !LIST;
    bf16x8 aA[RB][KS], aB[DB ? RB : 1][DB ? KS : 1];
    {
        const uint64_t prow0 = (uint64_t)blockIdx.x * UNIT + (uint64_t)wave * (RB * 32);
        set_lrow(prow0);
#pragma unroll
        for (int ks = 0; ks < KS; ks++) load_features(aA, prow0, ks);
    }
    uint32_t step = 0; // sub-slices consumed so far: buffer = step & 1
    stage(0, 0);
#ifdef LEANN_STAMPS
    uint64_t stW = 0, stBar = 0, stIss = 0, stN = 0, stCW = 0, stCG = 0, stCN = 0, stGL = 0, stGL1 = 0; // (SGPR budget: no finer split)
#endif
    f32x16 zero16;
#pragma unroll
    for (int i = 0; i < 16; i++) zero16[i] = 0.f;
    auto unit_body = [&](auto &a, auto &an, const uint64_t unit) __attribute__((always_inline)) {
        const uint64_t prow0 = unit * UNIT + (uint64_t)wave * (RB * 32);
        const uint64_t next_unit = unit + gridDim.x;
        const bool last_unit = next_unit >= n_units;
        const uint64_t nprow0 = (last_unit ? unit : next_unit) * UNIT + (uint64_t)wave * (RB * 32);
        set_lrow(nprow0); // long before the last G visit reads them
        float ssq[RB][16];
        f32x16 accA[RB], accB[RB]; // ping-pong accumulators of the column tiles; the finished one is squared into ssq
#pragma unroll
        for (int rb = 0; rb < RB; rb++) {
            accB[rb] = zero16;
#pragma unroll
            for (int i = 0; i < 16; i++) ssq[rb][i] = 0.f;
        }
        float inv[RB];
#pragma unroll
        for (int rb = 0; rb < RB; rb++) inv[rb] = 1.f;
        // One sub-slice visit.  KIND (compile time): 0 = weight sub-slice, 1 = score sub-slice, 2 = the unit's LAST score sub-slice,
        // whose loop also refills the feature fragments for the next unit.  The three kinds are separate instantiations called in
        // sequence (not branches inside one visit loop): the refilled fragments are then redefined once, at the end of the unit
        // body, and hipcc keeps them in their registers — as branches of one loop it gave the refilled values fresh registers, waited
        // for the just-issued loads right behind the loop and moved 2 x 128 AGPRs per unit to restore the loop-carried assignment.
        auto visit = [&](const int j, auto kind_c) __attribute__((always_inline)) {
            constexpr int KIND = decltype(kind_c)::value;
            const int buf = step & 1;
#ifdef LEANN_STAMPS
            const uint64_t t0 = __builtin_amdgcn_s_memtime();
#endif
            // my DMA pieces of sub-slice j have landed (issued a whole sub-slice ago) -> everybody's have, and everybody
            // has finished reading the other buffer -> it can take sub-slice j+1 (of this unit or of the next one)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef LEANN_STAMPS
            const uint64_t tW = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_s_barrier();
#ifdef LEANN_STAMPS
            const uint64_t tA = __builtin_amdgcn_s_memtime();
#endif
            // The next sub-slice (of this unit, or the first of the next one) is staged into the other buffer by LDS-DMA pieces
            // carried inside this visit's MFMA stream: wave w owns k-steps w, w + 4, w + 8, w + 12, four 1-KiB pieces each (a G
            // sub-slice has three; its fourth piece lands in the unused KiB of the k-step).  Staged on the last unit too
            // (harmless; drained before the kernel ends).
            const int jn = j + 1 < nsub ? j + 1 : 0;
            const bool nw = jn < nsw;
            const char *dbase = nw ? reinterpret_cast<const char *>(Wp) + (size_t)jn * (SUB * 32)
                                   : reinterpret_cast<const char *>(Gp) + (size_t)(jn - nsw) * (96 * 32);
            const uint32_t dstride = nw ? dp * 32 : (uint32_t)nqt * (96 * 32); // bytes between k-steps of the source image
            const char *dsb[4];
            uint32_t dld[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int ks = wave + NWV * t;
                dsb[t] = dbase + (size_t)ks * dstride;
                dld[t] = lds_base + (uint32_t)(buf ^ 1) * SUBB + ks * KSB;
            }
            if (j == 0) { // the unit's feature fragments have landed (vmcnt(0) above): pin them into their AGPRs here
#pragma unroll
                for (int rb = 0; rb < RB; rb++)
#pragma unroll
                    for (int ks = 0; ks < KS; ks++) asm volatile("" : "+a"(a[rb][ks]));
            }
            const uint32_t waddr = lds_base + (uint32_t)buf * SUBB + fro; // LDS byte address of this lane's fragment slot
#ifdef LEANN_STAMPS
            const uint64_t tB = __builtin_amdgcn_s_memtime();
#endif
            if constexpr (KIND == 0 || KIND == 3) {
                // 4 column tiles x KS k-steps, flat: fragment f+RING is requested while fragment f is multiplied.  The reads
                // are asm statements (the compiler's scheduler otherwise sinks every read to just before its use, which with
                // one wave per SIMD exposes the LDS latency 64 times per sub-slice); their lgkmcnt is counted by hand.
                static_assert(RB == 2 && RING == 4 && KS >= 3, "the per-k-step asm statements are written for two passage blocks and a 4-deep ring");
                bf16x8 bq[RING + 1];
                static_for<RING>([&bq, &waddr](auto fc) __attribute__((always_inline)) {
                    constexpr int f = decltype(fc)::value;
                    FSTAT_DS_READ(bq[f], waddr, (f % KS) * KSB + (f / KS) * 1024);
                });
                static_for<4 * KS>([&accA, &accB, &bq, &a, &an, &ssq, &waddr, &dsb, &dld, &goff, &nprow0, &load_feature1, &j](auto fc) __attribute__((always_inline)) {
                    constexpr int f = decltype(fc)::value, ct = f / KS, ks = f % KS, NF = 4 * KS;
                    f32x16 &c0 = (ct & 1) ? accB[0] : accA[0];
                    f32x16 &c1 = (ct & 1) ? accB[1] : accA[1];
                    // ring of RING + 1 registers: fragment f lives in slot f % (RING + 1); the read issued here (fragment
                    // f + RING) goes to the slot fragment f - 1 just left
                    bf16x8 &b = bq[f % (RING + 1)];
                    bf16x8 &nb = bq[(f + RING) % (RING + 1)];
                    constexpr int noff = ((f + RING) % KS) * KSB + ((f + RING) / KS) * 1024;
                    // squares of the previously finished tile: element ks - 2 of each block, two k-steps behind its last write
                    // (>= 4 MFMAs = 128 cycles, hipcc does not see the MFMA -> VALU hazard); the last step takes the remaining three
                    f32x16 &p0 = (ct & 1) ? accA[0] : accB[0];
                    f32x16 &p1 = (ct & 1) ? accA[1] : accB[1];
                    constexpr int e = ks >= 2 ? ks - 2 : 0;
                    constexpr bool dma = (f & 3) == 1;                    // piece f / 4 of the next sub-slice rides in this statement
                    constexpr int dt = (f >> 2) >> 2, doff = ((f >> 2) & 3) * 1024; // its k-step group and immediate offset
                    if constexpr (ks < 2) {
                        if constexpr (ks == 0) FSTAT_W_RDZ(c0, c1, a[0][ks], a[1][ks], b, nb, waddr, noff, RING - 1);
                        else if constexpr (dma) FSTAT_W_RD_D(c0, c1, a[0][ks], a[1][ks], b, nb, waddr, noff, RING - 1, goff, dsb[dt], dld[dt], doff);
                        else FSTAT_W_RD(c0, c1, a[0][ks], a[1][ks], b, nb, waddr, noff, RING - 1);
                    } else if constexpr (ks < KS - 1) {
                        if constexpr (f + RING < NF) {
                            if constexpr (dma) FSTAT_W_RD_SQ1_D(c0, c1, a[0][ks], a[1][ks], b, nb, waddr, noff, RING - 1, ssq[0][e], p0[e], ssq[1][e], p1[e], goff, dsb[dt], dld[dt], doff);
                            else FSTAT_W_RD_SQ1(c0, c1, a[0][ks], a[1][ks], b, nb, waddr, noff, RING - 1, ssq[0][e], p0[e], ssq[1][e], p1[e]);
                        } else {
                            if constexpr (dma) FSTAT_W_NR_SQ1_D(c0, c1, a[0][ks], a[1][ks], b, NF - f - 1, ssq[0][e], p0[e], ssq[1][e], p1[e], goff, dsb[dt], dld[dt], doff);
                            else FSTAT_W_NR_SQ1(c0, c1, a[0][ks], a[1][ks], b, NF - f - 1, ssq[0][e], p0[e], ssq[1][e], p1[e]);
                        }
                    } else {
                        if constexpr (f + RING < NF)
                            FSTAT_W_RD_SQ3(c0, c1, a[0][ks], a[1][ks], b, nb, waddr, noff, RING - 1, ssq[0][13], p0[13], ssq[1][13], p1[13],
                                           ssq[0][14], p0[14], ssq[1][14], p1[14], ssq[0][15], p0[15], ssq[1][15], p1[15]);
                        else
                            FSTAT_W_NR_SQ3(c0, c1, a[0][ks], a[1][ks], b, NF - f - 1, ssq[0][13], p0[13], ssq[1][13], p1[13], ssq[0][14], p0[14],
                                           ssq[1][14], p1[14], ssq[0][15], p0[15], ssq[1][15], p1[15]);
                    }
                    if constexpr (DB && KIND == 3 && f < RB * KS) // next unit's fragments: all issued in the first half of the visit,
                        load_feature1(an, nprow0, f / KS, f % KS); // so that the vmcnt(0) opening the next visit finds them landed
                });
                // the last MFMAs' results: hipcc does not see the MFMA -> VALU hazard (no hardware interlock); the operands pin every
                // compiler-generated reader or copy of the accumulators (loop-carried across visits) below the pad
                asm volatile("s_nop 7\n\ts_nop 7" : "+v"(accA[0]), "+v"(accA[1]), "+v"(accB[0]), "+v"(accB[1]));
#ifdef LEANN_STAMPS
                const uint64_t tN = __builtin_amdgcn_s_memtime();
                stCW += tN - tB;
#endif
                if (j == nsw - 1) { // all columns seen: flush the last tile, then row norms (candle.rs:218-225)
                    // The encode tiles are computed transposed (A = weight fragment, B = feature fragment: D[i = column][j = passage]),
                    // so a lane's 16 registers of every tile are 16 columns of ONE passage (j = lane & 31): the squares were summed
                    // per register slot by the fillers, the row norm is those 16 partial sums plus the other half wave's (columns
                    // i + 4) — one cross-lane exchange per block instead of a 5-step DPP tree over 32 values, and the result is
                    // already per lane-passage, the layout the score epilogue wants.
#pragma unroll
                    for (int rb = 0; rb < RB; rb++) {
                        float p[16];
#pragma unroll
                        for (int reg = 0; reg < 16; reg++) p[reg] = fmaf(accB[rb][reg], accB[rb][reg], ssq[rb][reg]);
#pragma unroll
                        for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
                            for (int reg = 0; reg < w; reg++) p[reg] = p[reg] + p[reg + w];
                        const float tot = p[0] + __shfl_xor(p[0], 32);
                        const float v = sqrtf(tot);
                        inv[rb] = 1.0f / (v < 1e-12f ? 1e-12f : v);
                    }
                }
            } else {
                const int qt = j - nsw;
                // last sub-slice of the unit (the last query tile): each feature register is refilled for the next unit right after its last
                // use; unconditional (the last unit re-reads its own rows) so that no branch or register copy sits between MFMAs
                f32x16 sc[RB];
                // (the ring's first reads sit inside each instantiation: with a branch between an asm read and the statement that
                // waits for it, hipcc copies the not-yet-landed destination registers at the join)
                auto g_loop = [&sc, &a, &an, &waddr, &nprow0, &load_features, &dsb, &dld, &goff](auto pfc) __attribute__((always_inline)) {
                    constexpr bool PF = decltype(pfc)::value;
                    bf16x8 gq[RING + 1];
                    static_for<RING>([&gq, &waddr](auto fc) __attribute__((always_inline)) {
                        constexpr int f = decltype(fc)::value;
                        FSTAT_DS_READ(gq[f], waddr, (f / 3) * KSB + (f % 3) * 1024);
                    });
                    // k-step major (pieces hi, lo, lo2 innermost): a feature fragment is dead after its k-step, so the next unit's
                    // feature loads spread over the whole visit instead of bunching in the last third
                    static_for<KS>([&sc, &gq, &a, &an, &waddr, &nprow0, &load_features, &dsb, &dld, &goff](auto kc) __attribute__((always_inline)) {
                        constexpr int ks = decltype(kc)::value, f = 3 * ks, NF = 3 * KS;
                        bf16x8 &s0 = gq[f % 5], &s1 = gq[(f + 1) % 5], &s2 = gq[(f + 2) % 5], &s4 = gq[(f + 4) % 5];
                        constexpr int o4 = ((f + 4) / 3) * KSB + ((f + 4) % 3) * 1024, o5 = ((f + 5) / 3) * KSB + ((f + 5) % 3) * 1024,
                                      o6 = ((f + 6) / 3) * KSB + ((f + 6) % 3) * 1024;
                        constexpr int dt = ks >> 2, doff = (ks & 3) * 1024; // LDS-DMA piece ks of the next sub-slice
                        if constexpr (f + 6 < NF) { // C[i = query][j = passage]
                            if constexpr (ks == 0)
                                FSTAT_GK_FULL_Z(sc[0], sc[1], s0, s1, s2, s4, a[0][ks], a[1][ks], waddr, o4, o5, o6, goff, dsb[dt], dld[dt], doff);
                            else
                                FSTAT_GK_FULL("%[x0]", "%[x1]", sc[0], sc[1], s0, s1, s2, s4, a[0][ks], a[1][ks], waddr, o4, o5, o6, goff, dsb[dt], dld[dt], doff);
                        } else if constexpr (f + 5 < NF) {
                            FSTAT_GK_T1(sc[0], sc[1], s0, s1, s2, s4, a[0][ks], a[1][ks], waddr, o4, o5, goff, dsb[dt], dld[dt], doff);
                        } else {
                            FSTAT_GK_T0(sc[0], sc[1], s0, s1, s2, a[0][ks], a[1][ks], waddr, goff, dsb[dt], dld[dt], doff);
                        }
                        // refill for the next unit, LEANN_FSTAT_PFD k-steps behind the fragment's last MFMA
                        if constexpr (PF && ks >= LEANN_FSTAT_PFD) load_features(an, nprow0, ks - LEANN_FSTAT_PFD);
                    });
                    if constexpr (PF)
                        static_for<LEANN_FSTAT_PFD>([&an, &nprow0, &load_features](auto kc) __attribute__((always_inline)) {
                            load_features(an, nprow0, KS - LEANN_FSTAT_PFD + decltype(kc)::value);
                        });
                };
                g_loop(std::integral_constant<bool, KIND == 2 && !DB>{});
                asm volatile("s_nop 7\n\ts_nop 7" : "+v"(sc[0]), "+v"(sc[1])); // as above: no reader of the score tiles above the pad
#ifdef LEANN_STAMPS
                if (qt == 0) stGL += __builtin_amdgcn_s_memtime() - tB; else stGL1 += __builtin_amdgcn_s_memtime() - tB;
#endif
                if (em.thr) {
                    // thresholds of this lane's 16 queries from the LDS copy (4 x ds_read_b128); one pass that only ORs compare
                    // masks, and a branch into the (rare) emission code per passage block instead of one per score
                    float th[16];
#pragma unroll
                    for (int g4 = 0; g4 < 4; g4++) {
                        const float4 t4 = *reinterpret_cast<const float4 *>(sThr + qt * 32 + 4 * lh + 8 * g4);
                        th[4 * g4] = t4.x; th[4 * g4 + 1] = t4.y; th[4 * g4 + 2] = t4.z; th[4 * g4 + 3] = t4.w;
                    }
#pragma unroll
                    for (int rb = 0; rb < RB; rb++) {
                        const uint64_t row = prow0 + rb * 32 + l31;
                        float sv[16];
                        // "does any score reach its threshold" as max(sv - th) >= 0: a - b >= 0 <=> a >= b in IEEE arithmetic (th = +-inf and
                        // NaN scores included: fmaxf drops NaNs), and 8 v_pk_mul + 8 v_pk_add + 8 v_max3 instead of 16 compares whose
                        // lane masks hipcc packed bit by bit through s_or / v_cndmask / shifts (~65 instructions per block)
                        float m = __uint_as_float(0xFF800000u);
#pragma unroll
                        for (int reg = 0; reg < 16; reg++) {
                            sv[reg] = sc[rb][reg] * inv[rb];
                            m = fmaxf(m, sv[reg] - th[reg]);
                        }
                        const bool any = m >= 0.0f;
                        if (any && row < n) {
                            const uint64_t pos = em.pos0 + row;
                            if (!em.allow || ((em.allow[pos >> 3] >> (pos & 7)) & 1)) {
#pragma unroll
                                for (int reg = 0; reg < 16; reg++) {
                                    if (sv[reg] >= th[reg]) {
                                        const uint32_t q = qt * 32 + 4 * lh + (reg & 3) + 8 * (reg >> 2);
                                        const uint32_t slot = atomicAdd(&em.cnt[q], 1u);
                                        if (slot < em.cap) em.list[(size_t)q * em.cap + slot] = ((uint64_t)(~f32_orderable(sv[reg])) << 32) | (uint32_t)pos;
                                    }
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int rb = 0; rb < RB; rb++) {
                        const uint64_t row = prow0 + rb * 32 + l31;
                        if (row < n) {
                            float *dstS = S + (size_t)(qt * 32 + 4 * lh) * n_rows_s + row;
#pragma unroll
                            for (int reg = 0; reg < 16; reg++) {
                                const uint32_t qo = (reg & 3) + 8 * (reg >> 2);
                                if (qt * 32 + 4 * lh + qo < nq) dstS[(size_t)qo * n_rows_s] = sc[rb][reg] * inv[rb];
                            }
                        }
                    }
                }
            }
#ifdef LEANN_STAMPS
            {
                asm volatile("s_nop 0" ::: "memory");
                const uint64_t tC = __builtin_amdgcn_s_memtime();
                // accumulated in SGPR-able scalars and flushed once at kernel end: an atomic here would sit in the vmcnt queue
                // that the next visit's wait measures
                stW += tW - t0; stBar += tA - tW; stIss += tB - tA; stN += 1;
                if (j >= nsw) stCG += tC - tB;
                else if (j == nsw - 1) stCN += tC - tB; // whole last W visit (its loop is also in stCW): norm tail = stCN - stCW share
            }
#endif
            step++;
        };
        if constexpr (DB) { // the first weight visit also refills the OTHER register set for the next unit
            visit(0, std::integral_constant<int, 3>{});
            for (int j = 1; j < nsw; j++) visit(j, std::integral_constant<int, 0>{});
        } else {
            for (int j = 0; j < nsw; j++) visit(j, std::integral_constant<int, 0>{});
        }
        for (int j = nsw; j < nsub - 1; j++) visit(j, std::integral_constant<int, 1>{});
        visit(nsub - 1, std::integral_constant<int, 2>{});
    };
    if constexpr (DB) {
        for (uint64_t unit = blockIdx.x; unit < n_units; unit += 2 * (uint64_t)gridDim.x) {
            unit_body(aA, aB, unit);
            if (unit + gridDim.x < n_units) unit_body(aB, aA, unit + gridDim.x);
        }
    } else {
        for (uint64_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) unit_body(aA, aA, unit);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the pieces staged during the last visit: no DMA into LDS after the workgroup ends
#ifdef LEANN_STAMPS
    if (tid == 0) {
        atomicAdd(&g_fstat_stamps[0], stW); atomicAdd(&g_fstat_stamps[1], stBar); atomicAdd(&g_fstat_stamps[2], stIss);
        atomicAdd(&g_fstat_stamps[4], stCW); atomicAdd(&g_fstat_stamps[5], stCG); atomicAdd(&g_fstat_stamps[6], stN);
        atomicAdd(&g_fstat_stamps[12], stCN); atomicAdd(&g_fstat_stamps[13], stGL); atomicAdd(&g_fstat_stamps[14], stGL1);
    }
#endif
}
