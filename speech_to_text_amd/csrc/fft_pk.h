// Packed-f32 forms of the register DFTs of fft_core.h for gfx950 device code.
//
// Why: measured on MI355X (tools/ubench), one wave issues at most one VALU instruction every ~4.4
// cycles, whatever its dependencies; a second wave on the SIMD fills the other half of the pipe for
// v_add/v_mul (2.2 cycles per wave instruction at saturation).  v_pk_{add,mul,fma}_f32 take the same
// 4.4-cycle slot but process a complex number (re, im in an even-aligned register pair) at once, so a
// kernel that runs two waves per SIMD spends half as long in its FFT phases; v_pk_fma_f32 also has
// twice the FMA throughput of v_fma_f32 at saturation.  The compiler emits the packed forms for
// float2 vector arithmetic but turns the operand swizzles (multiply by +-i, complex multiply, |.|^2
// cross terms) into v_mov / v_xor pairs, so those few forms are written with the instruction's own
// op_sel / neg operand modifiers.  Same arithmetic as fft_core.h (reference: torch.stft(n_fft=400)
// reached from src/data.py:338 via HF feature_extraction_whisper.py:149).
#pragma once

#include <hip/hip_runtime.h>

namespace sttfe {

typedef float f2 __attribute__((ext_vector_type(2)));

// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2 add_mi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + (+i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 add_pi(f2 a, f2 b) {
    f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// complex product (a.x b.x - a.y b.y, a.x b.y + a.y b.x) in two instructions
__device__ __forceinline__ f2 cmul_pk(f2 a, f2 b) {
    f2 t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(t) : "v"(a), "v"(b));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(b), "v"(t));
    return r;
}
// (s.x^2 + d.y^2, s.y^2 + d.x^2): the two power spectra of a packed pair of real frames
__device__ __forceinline__ f2 power_pair(f2 s, f2 d) {
    f2 m = s * s, r;
    asm("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[1,1,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(d), "v"(m));
    return r;
}

__device__ __forceinline__ void dft4_pk(f2& a0, f2& a1, f2& a2, f2& a3) {
    const f2 s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = add_mi(d02, d13);
    a3 = add_pi(d02, d13);
}

__device__ __forceinline__ void dft5_pk(f2& a0, f2& a1, f2& a2, f2& a3, f2& a4) {
    constexpr float C1 = 0.30901699437494742f;    // cos(2 pi / 5)
    constexpr float C2 = -0.80901699437494742f;   // cos(4 pi / 5)
    constexpr float S1 = 0.95105651629515357f;    // sin(2 pi / 5)
    constexpr float S2 = 0.58778525229247313f;    // sin(4 pi / 5)
    const f2 t1 = a1 + a4, t2 = a2 + a3, t3 = a1 - a4, t4 = a2 - a3;
    const f2 m1 = __builtin_elementwise_fma(f2{C2, C2}, t2, __builtin_elementwise_fma(f2{C1, C1}, t1, a0));
    const f2 m2 = __builtin_elementwise_fma(f2{C1, C1}, t2, __builtin_elementwise_fma(f2{C2, C2}, t1, a0));
    const f2 s1 = __builtin_elementwise_fma(f2{S2, S2}, t4, f2{S1, S1} * t3);
    const f2 s2 = __builtin_elementwise_fma(f2{-S1, -S1}, t4, f2{S2, S2} * t3);
    a0 = a0 + t1 + t2;
    a1 = add_mi(m1, s1);
    a4 = add_pi(m1, s1);
    a2 = add_mi(m2, s2);
    a3 = add_pi(m2, s2);
}

// 20-point DFT, Good-Thomas 4 x 5 (index maps as dft20 in fft_core.h): in -> out, natural order.
__device__ __forceinline__ void dft20_pk(const f2 (&in)[20], f2 (&out)[20]) {
    f2 t[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) t[i] = in[i];
#pragma unroll
    for (int b = 0; b < 5; ++b)
        dft4_pk(t[(4 * b) % 20], t[(5 + 4 * b) % 20], t[(10 + 4 * b) % 20], t[(15 + 4 * b) % 20]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        dft5_pk(t[(5 * c) % 20], t[(5 * c + 4) % 20], t[(5 * c + 8) % 20], t[(5 * c + 12) % 20],
                t[(5 * c + 16) % 20]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 5; ++d) out[(5 * c + 16 * d) % 20] = t[(5 * c + 4 * d) % 20];
}

// (The 16 x 32 transform of the Kaldi kernel stays on the scalar forms of fft_core.h: a packed version with
// constant twiddles in scalar register pairs was measured slower there, 0.166 vs 0.151 ms on config 2 -- its
// 8-wave workgroup already keeps two waves per SIMD in VALU phases, where packed adds gain nothing, and the
// compiler added ~360 v_mov to form the register pairs.)

}  // namespace sttfe
