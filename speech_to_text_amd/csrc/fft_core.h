// Register-resident small DFTs used by the STFT kernels (host + device).
//
// The 400-point transform of the Whisper front-end (reference: torch.stft(n_fft=400) reached from
// src/data.py:338 via HF feature_extraction_whisper.py:149) is factored 400 = 20 x 20
// (Cooley-Tukey across lanes through LDS), and each 20-point transform is a twiddle-free
// Good-Thomas 4 x 5 prime-factor DFT held entirely in registers.  The 512-point transform of the
// Kaldi front-end (HF audio_utils.py:988 with fft_length 512) is factored 16 x 32 with radix-4/2
// stages.  All functions are forward transforms, X[k] = sum_n x[n] exp(-2 pi i n k / N).
//
// This header is plain C++ so that tests/test_fft_host.py can compile the same code with g++ and
// check it against a naive O(N^2) DFT on the CPU.
#pragma once

#if defined(__HIPCC__)
#define STTFE_HD __host__ __device__ __forceinline__
#else
#define STTFE_HD inline
#endif

namespace sttfe {

struct cf {
    float x, y;
};

STTFE_HD cf operator+(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
STTFE_HD cf operator-(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
STTFE_HD cf cmul(cf a, cf b) { return cf{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
// a * (-i) and a * (+i)
STTFE_HD cf mul_mi(cf a) { return cf{a.y, -a.x}; }
STTFE_HD cf mul_pi(cf a) { return cf{-a.y, a.x}; }

STTFE_HD void dft2(cf& a0, cf& a1) {
    cf s = a0 + a1, d = a0 - a1;
    a0 = s;
    a1 = d;
}

STTFE_HD void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf s02 = a0 + a2, d02 = a0 - a2, s13 = a1 + a3, d13 = a1 - a3;
    a0 = s02 + s13;
    a2 = s02 - s13;
    a1 = d02 + mul_mi(d13);
    a3 = d02 + mul_pi(d13);
}

STTFE_HD void dft5(cf& a0, cf& a1, cf& a2, cf& a3, cf& a4) {
    constexpr float C1 = 0.30901699437494742f;    // cos(2 pi / 5)
    constexpr float C2 = -0.80901699437494742f;   // cos(4 pi / 5)
    constexpr float S1 = 0.95105651629515357f;    // sin(2 pi / 5)
    constexpr float S2 = 0.58778525229247313f;    // sin(4 pi / 5)
    cf t1 = a1 + a4, t2 = a2 + a3, t3 = a1 - a4, t4 = a2 - a3;
    cf m1{a0.x + C1 * t1.x + C2 * t2.x, a0.y + C1 * t1.y + C2 * t2.y};
    cf m2{a0.x + C2 * t1.x + C1 * t2.x, a0.y + C2 * t1.y + C1 * t2.y};
    cf s1{S1 * t3.x + S2 * t4.x, S1 * t3.y + S2 * t4.y};
    cf s2{S2 * t3.x - S1 * t4.x, S2 * t3.y - S1 * t4.y};
    a0 = a0 + t1 + t2;
    a1 = m1 + mul_mi(s1);
    a4 = m1 + mul_pi(s1);
    a2 = m2 + mul_mi(s2);
    a3 = m2 + mul_pi(s2);
}

// 20-point DFT, Good-Thomas: n = (5a + 4b) mod 20, k = (5c + 16d) mod 20.  in -> out (natural order).
STTFE_HD void dft20(const cf (&in)[20], cf (&out)[20]) {
    cf t[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) t[i] = in[i];
#pragma unroll
    for (int b = 0; b < 5; ++b)
        dft4(t[(4 * b) % 20], t[(5 + 4 * b) % 20], t[(10 + 4 * b) % 20], t[(15 + 4 * b) % 20]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        dft5(t[(5 * c) % 20], t[(5 * c + 4) % 20], t[(5 * c + 8) % 20], t[(5 * c + 12) % 20],
             t[(5 * c + 16) % 20]);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int d = 0; d < 5; ++d) out[(5 * c + 16 * d) % 20] = t[(5 * c + 4 * d) % 20];
}

// ---- power-of-two pieces for the 512-point (16 x 32) transform ----
constexpr float kSqrtHalf = 0.70710678118654752f;

// 8-point DFT in place, natural order in / out.
STTFE_HD void dft8(cf (&v)[8]) {
    // radix-2 decimation in frequency would bit-reverse; do 2 x 4 Cooley-Tukey explicitly.
    cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
    cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
    dft4(e0, e1, e2, e3);
    dft4(o0, o1, o2, o3);
    // twiddles W8^k on the odd half
    cf w1{(o1.x + o1.y) * kSqrtHalf, (o1.y - o1.x) * kSqrtHalf};     // * (1 - i)/sqrt2
    cf w2 = mul_mi(o2);
    cf w3{(o3.y - o3.x) * kSqrtHalf, -(o3.x + o3.y) * kSqrtHalf};    // * (-1 - i)/sqrt2
    v[0] = e0 + o0;
    v[4] = e0 - o0;
    v[1] = e1 + w1;
    v[5] = e1 - w1;
    v[2] = e2 + w2;
    v[6] = e2 - w2;
    v[3] = e3 + w3;
    v[7] = e3 - w3;
}

// 16-point DFT, natural order in -> out: 4 x 4 Cooley-Tukey (n = 4 n1 + n2, k = k1 + 4 k2).
STTFE_HD void dft16(const cf (&in)[16], cf (&out)[16]) {
    constexpr float C = 0.92387953251128674f;   // cos(pi/8)
    constexpr float S = 0.38268343236508977f;   // sin(pi/8)
    cf t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) t[i] = in[i];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(t[n2], t[4 + n2], t[8 + n2], t[12 + n2]);   // over n1 -> k1 at t[4 k1 + n2]
    // twiddle W16^(n2 k1)
    const cf w1{C, -S}, w2{kSqrtHalf, -kSqrtHalf}, w3{S, -C};
    t[4 * 1 + 1] = cmul(t[4 * 1 + 1], w1);
    t[4 * 1 + 2] = cmul(t[4 * 1 + 2], w2);
    t[4 * 1 + 3] = cmul(t[4 * 1 + 3], w3);
    t[4 * 2 + 1] = cmul(t[4 * 2 + 1], w2);
    t[4 * 2 + 2] = mul_mi(t[4 * 2 + 2]);
    t[4 * 2 + 3] = cmul(t[4 * 2 + 3], cf{-kSqrtHalf, -kSqrtHalf});
    t[4 * 3 + 1] = cmul(t[4 * 3 + 1], w3);
    t[4 * 3 + 2] = cmul(t[4 * 3 + 2], cf{-kSqrtHalf, -kSqrtHalf});
    t[4 * 3 + 3] = cmul(t[4 * 3 + 3], cf{-C, S});
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        dft4(t[4 * k1 + 0], t[4 * k1 + 1], t[4 * k1 + 2], t[4 * k1 + 3]);            // over n2 -> k2
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) out[k1 + 4 * k2] = t[4 * k1 + k2];
    }
}

// 32-point DFT, natural order in -> out: radix-2 split into two 16-point DFTs.
STTFE_HD void dft32(const cf (&in)[32], cf (&out)[32]) {
    cf e[16], o[16], E[16], O[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        e[i] = in[2 * i];
        o[i] = in[2 * i + 1];
    }
    dft16(e, E);
    dft16(o, O);
    // W32^k = exp(-2 pi i k / 32), k = 0..15
    constexpr float CW[16] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f,
                              0.70710678118654752f, 0.55557023301960218f, 0.38268343236508977f,
                              0.19509032201612825f, 0.0f, -0.19509032201612825f, -0.38268343236508977f,
                              -0.55557023301960218f, -0.70710678118654752f, -0.83146961230254524f,
                              -0.92387953251128674f, -0.98078528040323043f};
    constexpr float SW[16] = {0.0f, 0.19509032201612825f, 0.38268343236508977f, 0.55557023301960218f,
                              0.70710678118654752f, 0.83146961230254524f, 0.92387953251128674f,
                              0.98078528040323043f, 1.0f, 0.98078528040323043f, 0.92387953251128674f,
                              0.83146961230254524f, 0.70710678118654752f, 0.55557023301960218f,
                              0.38268343236508977f, 0.19509032201612825f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const cf t = (k == 0) ? O[0] : (k == 8 ? mul_mi(O[8]) : cmul(O[k], cf{CW[k], -SW[k]}));
        out[k] = E[k] + t;
        out[k + 16] = E[k] - t;
    }
}

}  // namespace sttfe
