// Shared device helpers for the gfx950 (MI355X / CDNA4) FlashAttention kernels.
//
// Everything here is written for wave64 + v_mfma_f32_32x32x16_{bf16,f16} only.
//
// Fragment conventions used by all three kernels (lane l: r = l & 31, h = l >> 5):
//   * MFMA A operand (32 rows x 16 k):  lane holds A[row r][k = 8h + j], j = 0..7
//   * MFMA B operand (16 k x 32 cols):  lane holds B[k = 8h + j][col r], j = 0..7
//   * MFMA C/D (32 x 32 fp32, 16 regs): reg i of lane = D[row (i&3) + 8(i>>2) + 4h][col r]
//   * "accumulator as next B operand": regs 8s..8s+7 of a C/D tile, rounded to
//     16 bit, are the B fragment of k-step s of a product that sums over the
//     tile's ROW index; element j then stands for row 16s + 8(j>>2) + 4h + (j&3),
//     and the A operand of that product is fetched in the same k order by two
//     ds_read_b64_tr_b16 (rows 16s + 4h + {0..3} and 16s + 8 + 4h + {0..3}).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace fa {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(8))) short i16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define FA_LDS __attribute__((address_space(3)))
#define FA_DEVINL __device__ __forceinline__

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// ---- dtype traits ---------------------------------------------------------
// kFoldScale: fold softmax_scale*log2(e) into the register-resident score operand (one extra rounding of it to
// 16 bit) so that the score MFMA chain, started from the row constant, delivers the exponent argument directly.
// bf16 only: there the extra rounding sits below the rounding of P itself (measured: relFro unchanged); at fp16 it
// would eat the margin of the reference's allclose(rtol 1e-2, atol 1e-3) criterion, so fp16 keeps the exact fma.
struct BF16 {
  typedef bf16x8 vec8;
  typedef __bf16 elem;
  static constexpr bool kFoldScale = true;
  static FA_DEVINL f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // VGPR-accumulator forms (mfma_v_* below): start a chain from c / from 0, continue it
  template <typename A4> static FA_DEVINL void mfma_v_first(f32x16& d, u32x4 a, A4 b, const f32x16& c);
  template <typename A4> static FA_DEVINL void mfma_v_first0(f32x16& d, u32x4 a, A4 b);
  template <typename A4> static FA_DEVINL void mfma_v_acc(f32x16& d, u32x4 a, A4 b);
};
struct FP16 {
  typedef f16x8 vec8;
  typedef _Float16 elem;
#ifndef FA_FP16_FOLD   // A/B hook (round 4): fp16 with the scale folded like bf16
#define FA_FP16_FOLD 0
#endif
  static constexpr bool kFoldScale = FA_FP16_FOLD != 0;
  static FA_DEVINL f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  template <typename A4> static FA_DEVINL void mfma_v_first(f32x16& d, u32x4 a, A4 b, const f32x16& c);
  template <typename A4> static FA_DEVINL void mfma_v_first0(f32x16& d, u32x4 a, A4 b);
  template <typename A4> static FA_DEVINL void mfma_v_acc(f32x16& d, u32x4 a, A4 b);
};

// MFMA with the accumulator pinned to ARCHITECTURAL VGPRs (inline asm); the B operand sits in VGPRs too unless
// FA_MFMA_B_AGPR is set (an A/B hook, OFF: hipcc then copies the operand back in front of every use).
// hipcc picks ONE register form for every MFMA builtin of a kernel: once a kernel needs AGPRs (one wave per SIMD, > 256
// registers) all its MFMA results land in AGPRs, and a result that VALU code consumes (scores -> exp) is then copied out
// element by element with v_accvgpr_read (measured: 2.6 extra vector instructions per MFMA in fa_bwd_dkv_v3.hip).  These
// forms keep such chains in VGPRs while the builtin MFMAs of the same kernel accumulate in AGPRs.
// What hipcc does not do for them (no hazard padding inside / after an asm statement): the caller guarantees that
//   * nothing reads or overwrites D within 12 wait states of the statement except the next MFMA of the same chain
//     (the accumulate chain itself needs none) -- the pipelines that use this consume D one block iteration later;
//   * A / B / C come from LDS reads or older VALU results (hipcc still inserts the s_waitcnt for loads it issued itself).
// tools/mfma_lint.py checks both on the built code objects (a CPU test).
typedef __attribute__((ext_vector_type(4))) unsigned agpr4_t;   // a 128-bit fragment living in a[N:N+3]
#ifndef FA_MFMA_B_AGPR
#define FA_MFMA_B_AGPR 0   // 1: the B operand of the mfma_v_* forms must live in AGPRs, 0: in VGPRs
#endif
#if FA_MFMA_B_AGPR
#define FA_MFMA_B(x) "a"(x)
#else
#define FA_MFMA_B(x) "v"(x)
#endif
#define FA_MFMA_ASM_(NAME, OP)                                                                                      \
  FA_DEVINL void NAME##_first(f32x16& d, u32x4 a, agpr4_t b, const f32x16& c) {                                    \
    asm volatile(OP " %0, %1, %2, %3" : "=&v"(d) : "v"(a), FA_MFMA_B(b), "v"(c));                                        \
  }                                                                                                                \
  FA_DEVINL void NAME##_first0(f32x16& d, u32x4 a, agpr4_t b) {                                                    \
    asm volatile(OP " %0, %1, %2, 0" : "=&v"(d) : "v"(a), FA_MFMA_B(b));                                                 \
  }                                                                                                                \
  FA_DEVINL void NAME##_acc(f32x16& d, u32x4 a, agpr4_t b) {                                                       \
    asm volatile(OP " %0, %1, %2, %0" : "+v"(d) : "v"(a), FA_MFMA_B(b));                                                 \
  }
FA_MFMA_ASM_(mfma_v_bf16, "v_mfma_f32_32x32x16_bf16")
FA_MFMA_ASM_(mfma_v_f16, "v_mfma_f32_32x32x16_f16")
#undef FA_MFMA_ASM_
// ---- a PINNED accumulator file (fa_bwd_dq_v4.hip) ---------------------------------------------------------------------
// One wave per SIMD owns 256 accumulator registers beside its 256 architectural ones, and everything a VGPR-form asm MFMA
// touches except its A / B operands must be architectural (C and D share one AGPR bit).  The kernel therefore keeps what
// does NOT need to be architectural in accumulator registers named LITERALLY in its asm statements:
//     dQ (fa_bwd_dq_v4.hip)                                        dK/dV (fa_bwd_dkv_v4.hip)
//     a[0:63]     dQ^T accumulators: 4 blocks of 16               a[0:127]    dV^T and dK^T accumulators: 8 blocks of 16
//     a[64:127]   resident B operands: Q^T, dO^T fragments        a[128:191]  resident B operands: K^T, V^T fragments
//     a[128:129]  LSE rows of the next pass (a global load        a[192:193]  a query tile's row constants (likewise)
//                 may land in an accumulator register)
//     a[194:255]  hipcc's
// Why literal names and not "a" operands: hipcc splits the live range of such an operand where it pleases and copies it back
// (v_accvgpr_write) right in front of the asm statement, whose MFMA then reads the register inside the write's wait states
// -- hipcc pads no hazard for an asm statement (seen: a wrong row block in the fp16 causal kernel); and left to allocate the
// accumulators itself it kept three copies of them for three code regions, 188 v_accvgpr_mov and 112 registers that the
// prefetch needs.  Registers hipcc never allocates cannot be copied.  What keeps it out: FA_PIN_CLOBBERS on the statements
// that bracket every long live range (each chain start, the prologue and epilogue statements), and what proves it stayed out:
// tools/mfma_lint.py rule R4 (a CPU test) reads the code objects and fails on any other instruction that touches a[0:193].
constexpr int kAccBase = 0, kPinBase = 64, kPfLse = 128, kPinEnd = 194;
#define FA_PIN_CLOBBERS \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", \
  "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", \
  "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", \
  "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", \
  "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", \
  "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", \
  "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", \
  "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", \
  "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", \
  "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", \
  "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", \
  "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", \
  "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", \
  "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", \
  "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", \
  "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", \
  "a192", "a193"
FA_DEVINL void pin_reserve() { asm volatile("" ::: FA_PIN_CLOBBERS); }
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression in the body
template <int N, typename F>
FA_DEVINL void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
// prefetch loads (the caller counts them itself: vmcnt; s_nop 4: a descriptor fresh from a spill lane, see dma16 below) and
// their read-back
template <int A>
FA_DEVINL void pf_load4(__amdgpu_buffer_rsrc_t r, int voff) {
  static_assert(A >= kPfLse && A < kPinEnd, "prefetch register range");
  asm volatile("s_nop 4\n\tbuffer_load_dword a[%c2], %0, %1, 0 offen" :: "v"(voff), "s"(r), "i"(A));
}
template <int A>
FA_DEVINL float acc_read1() {
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(A));
  return v;
}
template <int A>
FA_DEVINL u32x4 acc_read4() {   // a[A : A + 3]
  return u32x4{__builtin_bit_cast(unsigned, acc_read1<A>()), __builtin_bit_cast(unsigned, acc_read1<A + 1>()),
               __builtin_bit_cast(unsigned, acc_read1<A + 2>()), __builtin_bit_cast(unsigned, acc_read1<A + 3>())};
}
template <int A>
FA_DEVINL void acc_write4(u32x4 v) {   // a[A : A + 3]; the caller leaves the write -> MFMA wait states
  asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\t"
               "v_accvgpr_write_b32 a[%c7], %3"
               :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "i"(A), "i"(A + 1), "i"(A + 2), "i"(A + 3));
}
template <int F> FA_DEVINL void pin_write(u32x4 v) { static_assert(F >= 0 && F < 32, "pinned fragments: a[64 + 4F ..], 16 in the dQ kernel, 16..31 in the dK/dV kernel"); acc_write4<kPinBase + 4 * F>(v); }
template <int F> FA_DEVINL u32x4 pin_read() { return acc_read4<kPinBase + 4 * F>(); }
// a 16-register accumulator block a[A : A + 15]: zero, read out (after the last MFMA's 12+ wait states: the callers sit behind
// a barrier)
template <int A>
FA_DEVINL void acc_zero16() {
  static_for<4>([](auto q_) __attribute__((always_inline)) { acc_write4<A + 4 * decltype(q_)::value>(u32x4{0u, 0u, 0u, 0u}); });
}
template <int A>
FA_DEVINL f32x16 acc_read16() {
  f32x16 o;
  static_for<4>([&](auto q_) __attribute__((always_inline)) {
    constexpr int q = decltype(q_)::value;
    const u32x4 v = acc_read4<A + 4 * q>();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned w = v[j];   // (a copy: __builtin_bit_cast of the element expression itself reads element 0 every time)
      o[4 * q + j] = __builtin_bit_cast(float, w);
    }
  });
  return o;
}
// VGPR-form chain MFMAs with the B operand in pinned fragment F; `first` / `first0` carry the clobber list (one per chain
// start = two per block iteration is dense enough: nothing hipcc parks lives shorter than that)
#define FA_MFMA_ASM_PIN_(NAME, OP)                                                                                  \
  template <int F> FA_DEVINL void NAME##_first(f32x16& d, u32x4 a, const f32x16& c) {                              \
    asm volatile(OP " %0, %1, a[%c3:%c4], %2" : "=&v"(d) : "v"(a), "v"(c), "i"(kPinBase + 4 * F), "i"(kPinBase + 4 * F + 3) : FA_PIN_CLOBBERS); \
  }                                                                                                                \
  template <int F> FA_DEVINL void NAME##_first0(f32x16& d, u32x4 a) {                                              \
    asm volatile(OP " %0, %1, a[%c2:%c3], 0" : "=&v"(d) : "v"(a), "i"(kPinBase + 4 * F), "i"(kPinBase + 4 * F + 3) : FA_PIN_CLOBBERS); \
  }                                                                                                                \
  template <int F> FA_DEVINL void NAME##_acc(f32x16& d, u32x4 a) {                                                 \
    asm volatile(OP " %0, %1, a[%c2:%c3], %0" : "+v"(d) : "v"(a), "i"(kPinBase + 4 * F), "i"(kPinBase + 4 * F + 3)); \
  }                                                                                                                \
  /* a[A : A + 15] += a . b, both operands architectural: the accumulating MFMAs (their chain needs no wait states) */ \
  template <int A> FA_DEVINL void NAME##_into(u32x4 a, u32x4 b) {                                                  \
    asm volatile(OP " a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(a), "v"(b), "i"(A), "i"(A + 15));                    \
  }
FA_MFMA_ASM_PIN_(mfma_vp_bf16, "v_mfma_f32_32x32x16_bf16")
FA_MFMA_ASM_PIN_(mfma_vp_f16, "v_mfma_f32_32x32x16_f16")
#undef FA_MFMA_ASM_PIN_
struct MfmaPin {   // T-dispatch of the forms above
  template <int F> static FA_DEVINL void first(BF16*, f32x16& d, u32x4 a, const f32x16& c) { mfma_vp_bf16_first<F>(d, a, c); }
  template <int F> static FA_DEVINL void first0(BF16*, f32x16& d, u32x4 a) { mfma_vp_bf16_first0<F>(d, a); }
  template <int F> static FA_DEVINL void acc(BF16*, f32x16& d, u32x4 a) { mfma_vp_bf16_acc<F>(d, a); }
  template <int A> static FA_DEVINL void into(BF16*, u32x4 a, u32x4 b) { mfma_vp_bf16_into<A>(a, b); }
  template <int F> static FA_DEVINL void first(FP16*, f32x16& d, u32x4 a, const f32x16& c) { mfma_vp_f16_first<F>(d, a, c); }
  template <int F> static FA_DEVINL void first0(FP16*, f32x16& d, u32x4 a) { mfma_vp_f16_first0<F>(d, a); }
  template <int F> static FA_DEVINL void acc(FP16*, f32x16& d, u32x4 a) { mfma_vp_f16_acc<F>(d, a); }
  template <int A> static FA_DEVINL void into(FP16*, u32x4 a, u32x4 b) { mfma_vp_f16_into<A>(a, b); }
};
template <typename A4> FA_DEVINL void BF16::mfma_v_first(f32x16& d, u32x4 a, A4 b, const f32x16& c) { mfma_v_bf16_first(d, a, b, c); }
template <typename A4> FA_DEVINL void BF16::mfma_v_first0(f32x16& d, u32x4 a, A4 b) { mfma_v_bf16_first0(d, a, b); }
template <typename A4> FA_DEVINL void BF16::mfma_v_acc(f32x16& d, u32x4 a, A4 b) { mfma_v_bf16_acc(d, a, b); }
template <typename A4> FA_DEVINL void FP16::mfma_v_first(f32x16& d, u32x4 a, A4 b, const f32x16& c) { mfma_v_f16_first(d, a, b, c); }
template <typename A4> FA_DEVINL void FP16::mfma_v_first0(f32x16& d, u32x4 a, A4 b) { mfma_v_f16_first0(d, a, b); }
template <typename A4> FA_DEVINL void FP16::mfma_v_acc(f32x16& d, u32x4 a, A4 b) { mfma_v_f16_acc(d, a, b); }
// `x` stays in its registers up to this point (an empty asm that reads it).  A __device__ function on purpose: an asm
// statement with an AMDGPU register constraint written directly in a __global__ body makes the HOST pass drop the
// kernel's stub without a diagnostic (undefined symbol at load time).
FA_DEVINL void keep_live(const f32x16& x) { asm volatile("" ::"v"(x)); }
// `x` becomes a value DEFINED at this point of the chain of volatile asm statements: an instruction that consumes it cannot
// be placed above the asm MFMA written in front of this call.  sched_barrier(0) pins the machine scheduler only -- the
// instruction selector's own linearisation had moved the first exp of a block one MFMA slot up in fa_bwd_dkv_v3.hip, to 10
// wait states behind the chain's last MFMA where an 8-pass MFMA's result wants 12 (tools/mfma_lint.py rule R3).
// Not free: hipcc pads one s_nop between an asm statement and a VALU instruction that reads its output.
FA_DEVINL float here(float x) {
  asm volatile("" : "+v"(x));
  return x;
}
#ifdef FA_DKV3_SKEW
#define FA_SKEW_NOPS_1 "s_nop 15\n\t"
#define FA_SKEW_NOPS_2 "s_nop 15\n\ts_nop 15\n\t"
#define FA_SKEW_NOPS_4 "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
#define FA_SKEW_CAT_(n) FA_SKEW_NOPS_##n
#define FA_SKEW_CAT(n) FA_SKEW_CAT_(n)
FA_DEVINL void wave_skew(int wave) {
  asm volatile("s_cmp_lt_u32 %0, 1\n\ts_cbranch_scc1 L_skew%=\n\t" FA_SKEW_CAT(FA_DKV3_SKEW)
               "s_cmp_lt_u32 %0, 2\n\ts_cbranch_scc1 L_skew%=\n\t" FA_SKEW_CAT(FA_DKV3_SKEW)
               "s_cmp_lt_u32 %0, 3\n\ts_cbranch_scc1 L_skew%=\n\t" FA_SKEW_CAT(FA_DKV3_SKEW)
               "L_skew%=:" ::"s"(wave) : "scc");
}
#endif
// the 12+ wait states between an asm MFMA's result and its first VALU reader (outside the pipelines, where the reader
// follows at once); a __device__ function for the same host-pass reason as keep_live
FA_DEVINL void settle_mfma(f32x16& x) { asm volatile("s_nop 15" : "+v"(x)); }
FA_DEVINL void settle_mfma(f32x16& x, f32x16& y) { asm volatile("s_nop 15" : "+v"(x), "+v"(y)); }
// a 128-bit value moved into accumulator registers (explicitly: a value DEFINED in AGPRs needs no copy at its uses)
FA_DEVINL agpr4_t to_agpr(u32x4 v) {
#if !FA_MFMA_B_AGPR
  return v;
#endif
  agpr4_t o;
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o[0]) : "v"(v[0]));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o[1]) : "v"(v[1]));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o[2]) : "v"(v[2]));
  asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(o[3]) : "v"(v[3]));
  return o;
}

template <typename T>
FA_DEVINL typename T::vec8 as_vec8(u32x4 v) {
  return __builtin_bit_cast(typename T::vec8, v);
}

// regs 8s..8s+7 of a 32x32 fp32 tile -> one 16-bit B fragment (round to nearest even).
template <typename T, int S>
FA_DEVINL typename T::vec8 pack8(const f32x16& x) {
  typename T::vec8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (typename T::elem)x[8 * S + j];
  return o;
}

// two fp32 values -> one dword of two 16-bit values (round to nearest even): a single v_cvt_pk_* where it is written
template <typename T>
FA_DEVINL unsigned pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f32x2_;
  typedef __attribute__((ext_vector_type(2))) typename T::elem e2_;
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_){a, b}, e2_));
}

// 16-bit fragment times an fp32 constant, rounded back to 16 bit (used ONCE per workgroup on a register-resident
// operand: folding softmax_scale*log2(e) into it lets the score MFMA chain start from the row constant and
// deliver the exponent argument itself, so the per-element fma disappears from the hot loop).
template <typename T>
FA_DEVINL typename T::vec8 scale_frag(typename T::vec8 v, float c) {
  typename T::vec8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (typename T::elem)((float)v[j] * c);
  return o;
}

// ---- LDS image ------------------------------------------------------------
// A tile is [rows][D] 16-bit elements, row stride D*2 bytes, stored in 16-byte
// chunks whose index is XOR-swizzled with the row so that BOTH the ds_read_b128
// row reads of an A/B operand (16-lane groups reading 16 different rows at one
// chunk) and the ds_read_b64_tr_b16 transposed reads (a 32-lane half reading
// 4 consecutive rows x 64 B) are bank-conflict free (banks = (addr/4) % 64).
//   D = 64  (128-B rows, two rows per 256-B bank window):
//       chunk ^= (bit1(row) << 2) | (bit3(row) << 1) | bit2(row)
//   D = 128 (256-B rows): chunk ^= ((row & 3) << 2) | ((row >> 2) & 3)
template <int D>
FA_DEVINL int swz_chunk(int row, int chunk) {
  if constexpr (D == 64) {
    return chunk ^ ((((row >> 1) & 1) << 2) | ((row >> 2) & 3));
  } else {
    static_assert(D == 128, "head dim must be 64 or 128");
    return chunk ^ (((row & 3) << 2) | ((row >> 2) & 3));
  }
}
template <int D>
FA_DEVINL int lds_off(int row, int chunk) {
  return row * (D * 2) + swz_chunk<D>(row, chunk) * 16;
}

// A value hipcc must treat as freshly computed here: address arithmetic that depends on it can no longer be hoisted
// to kernel entry and carried (or spilled) across the tile loop.
FA_DEVINL int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

// Lane index 0..63 recomputed from EXEC (needs all lanes active) instead of read from the thread-id VGPR: nothing then
// has to stay live (or be spilled) across a tile loop just to re-derive lane coordinates for the next pass.
FA_DEVINL int lane_id_now() {
  int x;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(x));
  return x;
}

FA_DEVINL u32x4 lds_read16(const FA_LDS char* p) { return *(const FA_LDS u32x4*)p; }
FA_DEVINL void lds_write16(FA_LDS char* p, u32x4 v) { *(FA_LDS u32x4*)p = v; }
FA_DEVINL void lds_write8(FA_LDS char* p, u32x2 v) { *(FA_LDS u32x2*)p = v; }

// Zero the first `bytes` of the workgroup's LDS (ragged last tiles: an out-of-range LDS-DMA may leave its destination
// untouched, and whatever lies there is multiplied by P = 0).  Thread id and the zero vector are made opaque so that
// this rarely taken block adds nothing to the register pressure of the kernel around it.
FA_DEVINL void lds_zero_fill(FA_LDS char* smem, int bytes, int nthreads, int tid) {
  const int t = opaque(tid);
  u32x4 z = {0u, 0u, 0u, 0u};
  asm volatile("" : "+v"(z));
  for (int i = t * 16; i < bytes; i += nthreads * 16) lds_write16(smem + i, z);
}

// ds_read_b64_tr_b16: needs EXEC all ones and an 8-byte aligned address per lane.
FA_DEVINL i16x4 lds_read_tr(const FA_LDS char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS i16x4*)p);
}

// A-operand fragment fetched TRANSPOSED from a row-major [row][D] image: result
// lane (r, h) element j = tile[row0 + 8(j>>2) + 4h + (j&3)][col0 + r].
// `a0` / `a1` are this lane's byte addresses for the two 4-row blocks (see tr_lane_off).
template <typename T>
FA_DEVINL typename T::vec8 lds_read_tr_frag(const FA_LDS char* a0, const FA_LDS char* a1) {
  i16x4 lo = lds_read_tr(a0);
  i16x4 hi = lds_read_tr(a1);
  i16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(typename T::vec8, v);
}

// Byte offset this lane supplies to ds_read_b64_tr_b16 for the 4-row block whose first
// row is `row0 + 4h` (row0 a multiple of 8) and whose 32 columns start at col32*32.
// Lane l: 16-lane group g = l >> 4 (h = g >> 1 picks rows, g & 1 picks the 16-column
// half), i = l & 15 -> row i >> 2, columns 4(i & 3) .. 4(i & 3) + 3.
template <int D>
FA_DEVINL int tr_lane_off(int lane, int row0, int col32) {
  const int g = lane >> 4, i = lane & 15;
  const int row = row0 + 4 * (g >> 1) + (i >> 2);
  const int p = i & 3;
  const int chunk = col32 * 4 + (g & 1) * 2 + (p >> 1);
  return lds_off<D>(row, chunk) + (p & 1) * 8;
}

// ---- global memory through buffer descriptors (hardware bounds check) -----
// Out-of-range loads return 0 and out-of-range stores are dropped, which is what
// gives ragged sequence lengths their masked tails without any branch.
FA_DEVINL __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
FA_DEVINL u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
}
FA_DEVINL float buf_load_f32(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
FA_DEVINL void buf_store16(__amdgpu_buffer_rsrc_t r, int off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
}
FA_DEVINL void buf_store_f32(__amdgpu_buffer_rsrc_t r, int off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}

// A sequence with queries but no keys (only a variable-length launch can present one; the fixed-length ABI rejects
// S_k = 0) has nothing to attend to: the forward defines O = 0 and LSE = -inf for its rows, the backward dQ = 0 and delta = 0.
// No special path: its K / V descriptors have size 0 (view_bytes below), so every K / V access is out of range -- loads
// return 0, LDS-DMA fetches nothing -- and the kernels' normalisation treats an empty row sum as 0.
// byte size of a K / V view of `rows` rows for its buffer descriptor (0 rows -> 0 bytes: every access is out of range)
// (min / max instead of a select: hipcc lowers the select to v_cndmask, which would put the descriptor word in a VGPR)
FA_DEVINL unsigned view_bytes(int rows, int rs, int rowb) {
  return (unsigned)(max(rows, 1) - 1) * rs + (unsigned)(min(rows, 1) * rowb);
}

// (Wait states inside the strings below: an M0 write needs one before the LDS-DMA that uses it, and a descriptor or scalar
// offset that hipcc has just restored from a spill lane (v_readlane: a VALU write of an SGPR) needs five before a buffer
// instruction reads it -- hipcc pads neither in front of an asm statement.  s_mov m0 + s_nop 3 covers both; a kernel with
// many descriptors and 100+ SGPRs does spill them, and a stale descriptor is a memory fault, not a wrong number.)
// ---- LDS-DMA (buffer_load_dwordx4 ... lds): 64 lanes x 16 B from per-lane global offsets `voff`
// (+ wave-uniform `soff`) to the 1 KiB of LDS starting at byte address `lds_addr` (wave-uniform).
// Inline asm on purpose: the builtin form makes hipcc drain vmcnt(0) before any later LDS read it
// cannot disambiguate (every ds_read_b64_tr_b16) and at every memory clobber, which serialises the
// prefetch ring.  The compiler does not count these loads: the kernels retire them with explicit
// counted s_waitcnt vmcnt(N) + s_barrier.  M0 (the DMA's LDS base) is saved and restored.
FA_DEVINL void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, int voff, int soff) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff));
}
// N consecutive 1-KiB pieces (N = 1, 2 or 4) of ONE matrix with M0 written once: piece i lands at lds_addr + 1024*i.
// The instruction's immediate offset moves the LDS destination AND the global address, so `voff[i]` must already be
// the piece's source offset MINUS 1024*i (callers precompute it; it stays >= 0 because piece i starts at tile row
// i * (1024 / row bytes) and rows are at least that long).  M0 is not restored: hipcc reserves it and never touches it in
// these kernels (no compiler-generated m0 use in the .s), which saves two scalar moves per piece on the scalar pipe
// that every wave of the SIMD shares.
template <int N>
FA_DEVINL void dma_pieces(__amdgpu_buffer_rsrc_t rsrc, unsigned lds_addr, const int* voff, int soff) {
  static_assert(N == 1 || N == 2 || N == 4, "1, 2 or 4 pieces per M0 setting (12-bit immediate offset)");
  if constexpr (N == 1) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff[0]), "s"(rsrc), "s"(soff));
  } else if constexpr (N == 2) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen offset:1024 lds"
                 :: "s"(lds_addr), "v"(voff[0]), "v"(voff[1]), "s"(rsrc), "s"(soff));
  } else {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 3\n\tbuffer_load_dwordx4 %1, %5, %6 offen lds\n\t"
                 "buffer_load_dwordx4 %2, %5, %6 offen offset:1024 lds\n\t"
                 "buffer_load_dwordx4 %3, %5, %6 offen offset:2048 lds\n\t"
                 "buffer_load_dwordx4 %4, %5, %6 offen offset:3072 lds"
                 :: "s"(lds_addr), "v"(voff[0]), "v"(voff[1]), "v"(voff[2]), "v"(voff[3]), "s"(rsrc), "s"(soff));
  }
}
FA_DEVINL unsigned lds_addr_of(const FA_LDS char* p) { return (unsigned)(uintptr_t)p; }
FA_DEVINL const FA_LDS char* lds_at(int addr) { return (const FA_LDS char*)(uintptr_t)(unsigned)addr; }   // absolute LDS byte address -> pointer

// ---- cross-half exchange (lanes l <-> l + 32) -----------------------------
// v_permlane32_swap_b32 vdst, src swaps lanes 32..63 of vdst with lanes 0..31 of src.  Fed two
// copies of v it leaves a = {low half's values in both halves}, b = {high half's values}.
// Inline asm on purpose: hipcc (ROCm 7.2) folds the two results of
// __builtin_amdgcn_permlane32_swap into one when they feed a commutative op (seen in the IR:
// r[0] + r[1] became r[0] + r[0]), which silently breaks every reduction below.
FA_DEVINL void swap_halves(float& a, float& b) {
  // 2 wait states between a VALU write of an operand and the permlane read (s_nop 1)
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
FA_DEVINL float half_max(float v) {
  float a = v, b = v;
  swap_halves(a, b);
  return __builtin_fmaxf(a, b);
}
FA_DEVINL float half_sum(float v) {
  float a = v, b = v;
  swap_halves(a, b);
  return a + b;
}

// ---- attention dropout (reference text Phase_6.md:54-113: "Philox lets forward and backward regenerate
// the same mask from (seed, offset) without storing it") --------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; the generator PyTorch / cuRAND / hipRAND use): counter-based, so the keep
// decision of attention weight (b, h, q, k) is a pure function of its coordinates and (seed, offset) -- the three kernels
// regenerate identical masks although they tile and orient the score matrix differently.
// One call yields 16 random bytes = the 4 x 4 patch of weights  q in [4*qg, 4*qg+4) x k in [4*kg, 4*kg+4):
// byte (k & 3) of word (q & 3).  A weight is KEPT iff its byte >= thresh (thresh = round(256 p), p quantised to 1/256)
// and kept weights are scaled by rp = 256 / (256 - thresh).  In every kernel a lane's 16 accumulator registers of a
// 32 x 32 block are four 1 x 4 (forward, dQ: lane = query) or 4 x 1 (dK/dV: lane = key) strips of four different
// patches g = 0..3 -- and the four lanes of a quad (lanes 4n..4n+3: four consecutive queries, or keys) hold strips of the
// SAME four patches.  Lane j of the quad therefore generates patch g = j only and the quad exchanges the words with DPP
// quad_perm broadcasts (quad_bcast): one Philox call per block and lane instead of four.  v_mul_hi / v_mul_lo are quarter
// rate on CDNA4, so even that call is most of the dropout kernels' extra time; the plain kernels are separate template
// instances and pay nothing.
struct Dropout {
  unsigned thresh;            // 0 = no dropout (the plain kernels are launched)
  unsigned seed_lo, seed_hi;  // Philox key
  unsigned offset;            // fourth counter word (the caller's Philox offset)
  float rp;                   // 1 / (1 - thresh / 256)
};
FA_DEVINL u32x4 philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
  constexpr unsigned kM0 = 0xD2511F53u, kM1 = 0xCD9E8D57u, kW0 = 0x9E3779B9u, kW1 = 0xBB67AE85u;
#pragma unroll
  for (int round = 0; round < 10; ++round) {
    // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of v_mul_hi + v_mul_lo, all quarter rate
    const unsigned long long p0 = (unsigned long long)kM0 * c0, p1 = (unsigned long long)kM1 * c2;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0;
    const unsigned hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    // three-input xor in one instruction (v_bitop3_b32, truth table 0x96); hipcc emits two v_xor_b32 for hi ^ c ^ k
    const unsigned n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
    c0 = n0;
    c1 = lo1;
    c2 = n2;
    c3 = lo0;
    k0 += kW0;
    k1 += kW1;
  }
  return u32x4{c0, c1, c2, c3};
}
// the 4 x 4 patch of random bytes holding weights (4*qg.., 4*kg..) of (batch*head) slice `bh`
FA_DEVINL u32x4 dropout_patch(const Dropout& d, int qg, int kg, int bh) {
  return philox4x32_10((unsigned)qg, (unsigned)kg, (unsigned)bh, d.offset, d.seed_lo, d.seed_hi);
}
// value of `v` in lane G of this lane's quad (all four lanes of the quad must be active: EXEC is full in these kernels)
template <int G>
FA_DEVINL unsigned quad_bcast(unsigned v) {
  // bound_ctrl: every lane of a quad_perm has a valid source, so `old` is never used -- saying so spares the v_mov that
  // initialises it and lets the DPP operand fold into the consuming instruction
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, G * 0x55 /* quad_perm:[G,G,G,G] */, 0xF, 0xF, true);
}
template <int G>
FA_DEVINL u32x4 quad_bcast4(const u32x4& v) {
  return u32x4{quad_bcast<G>(v[0]), quad_bcast<G>(v[1]), quad_bcast<G>(v[2]), quad_bcast<G>(v[3])};
}
FA_DEVINL unsigned select_word(const u32x4& w, int idx) {  // idx in 0..3, lane dependent
  const unsigned lo = (idx & 1) ? w[1] : w[0], hi = (idx & 1) ? w[3] : w[2];
  return (idx & 2) ? hi : lo;
}

// ---- workgroup -> work item, XCD aware ------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).
// Give every XCD one contiguous slice of the work list so that the q/k tiles of one
// (batch, head) stream their K/V (or Q/dO) through ONE L2.  Bijective for any n.
FA_DEVINL int xcd_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, x = b & 7, idx = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + idx;
}

// ---- epilogue: a wave's 32 x D transposed accumulator -> row-major global ---
// acc[db][i] holds OUT[row = lane & 31][col = db*32 + (i&3) + 8(i>>2) + 4h] * mul.
// The wave stages its tile in its own LDS area (32 rows x D*2 bytes, swizzled) and
// writes whole rows back with 16-byte stores (8 rows per 1-KiB wave instruction).
// `row0_bytes` = byte offset of the tile's first row in `dst`, `ors` = byte stride between output rows (D*2 for the
// reference's contiguous [B, H, S, D] outputs, H*D*2 for packed varlen rows).
template <int D, typename T>
FA_DEVINL void store_tile_rows(const f32x16 (&acc)[D / 32], float mul, FA_LDS char* stage,
                               __amdgpu_buffer_rsrc_t dst, int row0_bytes, int lane, int ors = D * 2) {
  // Opaque lane id: everything below is address arithmetic on `lane` that does not change from pass to pass, so hipcc
  // hoists it to kernel entry, keeps ~20 values live across the whole tile loop and, at the 168-register budget of
  // three workgroups per CU, spills them (22 dwords of scratch per lane in the headline dQ kernel).  Recomputing
  // them here costs ~30 VALU ops per pass.
  asm volatile("" : "+v"(lane));
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int db = 0; db < D / 32; ++db) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      typedef __attribute__((ext_vector_type(4))) typename T::elem e4;
      e4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = (typename T::elem)(acc[db][4 * g + j] * mul);
      lds_write8(stage + lds_off<D>(r, db * 4 + g) + 8 * h, __builtin_bit_cast(u32x2, v));
    }
  }
  // same wave wrote and reads: only the LDS counter needs to drain (wave-local ordering)
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
  constexpr int CPR = D / 8;
#pragma unroll
  for (int i = 0; i < (32 * CPR) / 64; ++i) {
    const int id = lane + 64 * i, row = id / CPR, c = id % CPR;
    u32x4 v = lds_read16(stage + lds_off<D>(row, c));
    buf_store16(dst, row0_bytes + row * ors + c * 16, v);
  }
}

}  // namespace fa
