// Host check of speech_to_text_amd/csrc/fft_core.h and of the 20x20 lane choreography used by the
// Whisper STFT kernel, against a naive O(N^2) double-precision DFT.  Built and run by
// tests/test_fft_host.py with g++ (no GPU needed).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <complex>
#include "../../speech_to_text_amd/csrc/fft_core.h"

using namespace sttfe;
typedef std::complex<double> cd;

static double frand() { return (double)rand() / RAND_MAX * 2.0 - 1.0; }

template <int N>
static double check_small(void (*f)(const cf (&)[N], cf (&)[N])) {
    cf in[N], out[N];
    for (int i = 0; i < N; ++i) in[i] = cf{(float)frand(), (float)frand()};
    f(in, out);
    double worst = 0;
    for (int k = 0; k < N; ++k) {
        cd acc = 0;
        for (int n = 0; n < N; ++n) acc += cd(in[n].x, in[n].y) * std::polar(1.0, -2.0 * M_PI * n * k / N);
        worst = std::fmax(worst, std::abs(acc - cd(out[k].x, out[k].y)));
    }
    return worst;
}

static void dft8_wrap(const cf (&in)[8], cf (&out)[8]) {
    cf v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[i];
    dft8(v);
    for (int i = 0; i < 8; ++i) out[i] = v[i];
}

int main() {
    srand(1234);
    double e20 = check_small<20>(dft20), e16 = check_small<16>(dft16), e8 = check_small<8>(dft8_wrap);
    double e32 = check_small<32>(dft32);
    printf("dft32 %.3e\n", e32);
    if (e32 > 3e-5) return 1;
    printf("dft20 %.3e dft16 %.3e dft8 %.3e\n", e20, e16, e8);
    if (e20 > 2e-5 || e16 > 2e-5 || e8 > 1e-5) return 1;

    // ---- 400-point, two real frames packed as one complex sequence, 20 x 20 choreography ----
    std::vector<float> fa(400), fb(400);
    for (int i = 0; i < 400; ++i) { fa[i] = (float)frand(); fb[i] = (float)frand(); }
    std::vector<cf> tw(400);
    for (int m = 0; m < 400; ++m) tw[m] = cf{(float)std::cos(2 * M_PI * m / 400), (float)-std::sin(2 * M_PI * m / 400)};
    static cf scratch[20][20];                 // [k1][n2]
    for (int j = 0; j < 20; ++j) {             // "lane" j = n2 : 20-pt DFT over n1 of z[20 n1 + j]
        cf in[20], out[20];
        for (int n1 = 0; n1 < 20; ++n1) in[n1] = cf{fa[20 * n1 + j], fb[20 * n1 + j]};
        dft20(in, out);
        for (int k1 = 0; k1 < 20; ++k1) scratch[k1][j] = cmul(out[k1], tw[(j * k1) % 400]);
    }
    static cf Z[20][20];                       // lane j = k1 holds Z[k1 + 20 k2] at [j][k2]
    for (int j = 0; j < 20; ++j) {
        cf in[20], out[20];
        for (int n2 = 0; n2 < 20; ++n2) in[n2] = scratch[j][n2];
        dft20(in, out);
        for (int k2 = 0; k2 < 20; ++k2) Z[j][k2] = out[k2];
    }
    double worst = 0, scale = 0;
    for (int j = 0; j < 20; ++j) {
        int partner = (20 - j) % 20;
        int nb = (j == 0) ? 11 : 10;
        for (int k2 = 0; k2 < nb; ++k2) {
            int k = j + 20 * k2;
            cf z = Z[j][k2];
            cf zp = (j == 0) ? (k2 == 0 ? Z[0][0] : Z[0][20 - k2]) : Z[partner][19 - k2];
            double pa = 0.25 * ((double)(z.x + zp.x) * (z.x + zp.x) + (double)(z.y - zp.y) * (z.y - zp.y));
            double pb = 0.25 * ((double)(z.y + zp.y) * (z.y + zp.y) + (double)(z.x - zp.x) * (z.x - zp.x));
            cd ra = 0, rb = 0;
            for (int n = 0; n < 400; ++n) {
                cd w = std::polar(1.0, -2.0 * M_PI * n * k / 400);
                ra += (double)fa[n] * w;
                rb += (double)fb[n] * w;
            }
            worst = std::fmax(worst, std::fabs(pa - std::norm(ra)));
            worst = std::fmax(worst, std::fabs(pb - std::norm(rb)));
            scale = std::fmax(scale, std::norm(ra));
        }
    }
    printf("stft400 power worst abs err %.3e (max power %.3e)\n", worst, scale);
    if (worst > 1e-4 * scale) return 2;
    printf("OK\n");
    return 0;
}
