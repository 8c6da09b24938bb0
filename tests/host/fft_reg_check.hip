// Host check of cheetah_amd/csrc/chx_fft_reg.h: every register FFT against a direct O(n^2) DFT in long double.
// Built and run by tests/test_abi_and_host.py (hipcc --cuda-host-only: no GPU, no HIP runtime call).
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "chx_fft_reg.h"

template <typename T, int R, bool INV, bool UPPER_ZERO>
static double check() {
    vec2<T> x[R];
    long double re[R], im[R];
    for (int i = 0; i < R; ++i) {
        const bool zero = UPPER_ZERO && i >= R / 2;
        re[i] = zero ? 0.0L : (long double)(T)(rand() / (double)RAND_MAX - 0.5);
        im[i] = zero ? 0.0L : (long double)(T)(rand() / (double)RAND_MAX - 0.5);
        x[i] = vec2<T>{(T)re[i], (T)im[i]};
    }
    if constexpr (R == 16 && UPPER_ZERO) chx_fft::fft16<T, INV, true>(x);
    else chx_fft::fft_small<T, R, INV>(x);
    double worst = 0.0;
    const long double pi = acosl(-1.0L);
    for (int k = 0; k < R; ++k) {
        long double sr = 0, si = 0;
        for (int j = 0; j < R; ++j) {
            const long double ph = (INV ? 2.0L : -2.0L) * pi * (long double)((j * k) % R) / R;
            sr += re[j] * cosl(ph) - im[j] * sinl(ph);
            si += re[j] * sinl(ph) + im[j] * cosl(ph);
        }
        worst = fmax(worst, fmax(fabs((double)(sr - x[k].x)), fabs((double)(si - x[k].y))));
    }
    return worst;
}

template <typename T>
static int run(const char* name, double tol) {
    int bad = 0;
    for (int rep = 0; rep < 50; ++rep) {
        const double e[] = {check<T, 2, false, false>(), check<T, 2, true, false>(), check<T, 4, false, false>(),
                            check<T, 4, true, false>(),  check<T, 8, false, false>(), check<T, 8, true, false>(),
                            check<T, 16, false, false>(), check<T, 16, true, false>(), check<T, 16, false, true>(),
                            check<T, 16, true, true>()};
        for (double v : e) bad += !(v < tol);
    }
    printf("%s: %s\n", name, bad ? "FAIL" : "ok");
    return bad;
}

int main() {
    srand(7);
    const int bad = run<float>("f32", 4e-6) + run<double>("f64", 8e-15);
    return bad ? 1 : 0;
}
