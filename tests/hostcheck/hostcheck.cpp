// TEST-ONLY host build of dicp_amd/csrc/dicp_math.h (g++, no GPU).
// Lets the CPU test-suite check the closed-form forward/backward formulas that the
// HIP kernels execute against the oracle's autograd.  Never loaded by dicp_amd.
#include "../../dicp_amd/csrc/dicp_math.h"

using namespace dicp;

template <typename T>
static void forward_t(const WeightParams& P, int n, int c, const T* src, const T* tgt, const int* idx,
                      const T* C, const T* r, const T* w0, T* acc, T* w_out) {
    for (int k = 0; k < NACC; ++k) acc[k] = T(0);
    const T zero3[3] = {T(0), T(0), T(0)};
    for (int i = 0; i < n; ++i) {
        const T* y = tgt + (long)idx[i] * c;
        PointState<T> s;
        if (P.mode == MODE_PT2PL) point_forward<T, MODE_PT2PL>(P, C, r, src + 3 * i, y, y + 3, w0[i], acc, s);
        else                      point_forward<T, MODE_PT2PT>(P, C, r, src + 3 * i, y, zero3, w0[i], acc, s);
        w_out[i] = s.w;
    }
}

template <typename T>
static void backward_t(const WeightParams& P, int n, int c, const T* src, const T* tgt, const int* idx,
                       const T* C, const T* r, const T* w0, const T* Gs, const T* gb,
                       T* gsrc, T* gtgt, T* gw0, T* gC, T* gr) {
    const T zero3[3] = {T(0), T(0), T(0)};
    for (int i = 0; i < n; ++i) {
        const T* y = tgt + (long)idx[i] * c;
        T gp[3], gy[3], gn[3], gw;
        if (P.mode == MODE_PT2PL) point_backward<T, MODE_PT2PL>(P, C, r, src + 3 * i, y, y + 3, w0[i], Gs, gb, gp, gy, gn, gw, gC, gr);
        else                      point_backward<T, MODE_PT2PT>(P, C, r, src + 3 * i, y, zero3, w0[i], Gs, gb, gp, gy, gn, gw, gC, gr);
        for (int k = 0; k < 3; ++k) { gsrc[3 * i + k] += gp[k]; gtgt[(long)idx[i] * c + k] += gy[k]; }
        if (c == 6) for (int k = 0; k < 3; ++k) gtgt[(long)idx[i] * c + 3 + k] += gn[k];
        gw0[i] += gw;
    }
}

extern "C" {

void hc_forward_f64(const WeightParams* P, int n, int c, const double* src, const double* tgt, const int* idx,
                    const double* C, const double* r, const double* w0, double* acc, double* w_out) {
    forward_t<double>(*P, n, c, src, tgt, idx, C, r, w0, acc, w_out);
}
void hc_forward_f32(const WeightParams* P, int n, int c, const float* src, const float* tgt, const int* idx,
                    const float* C, const float* r, const float* w0, float* acc, float* w_out) {
    forward_t<float>(*P, n, c, src, tgt, idx, C, r, w0, acc, w_out);
}
void hc_backward_f64(const WeightParams* P, int n, int c, const double* src, const double* tgt, const int* idx,
                     const double* C, const double* r, const double* w0, const double* Gs, const double* gb,
                     double* gsrc, double* gtgt, double* gw0, double* gC, double* gr) {
    backward_t<double>(*P, n, c, src, tgt, idx, C, r, w0, Gs, gb, gsrc, gtgt, gw0, gC, gr);
}
void hc_step_forward(const double* acc, int dim, const double* C, const double* r,
                     double* delta6, double* Cn, double* rn, double* Areg) {
    double A6[36];
    unpack_sym6(acc + ACC_A, A6);
    step_forward(A6, acc + ACC_B, dim, C, r, delta6, Cn, rn, Areg);
}
void hc_step_backward(const double* gCn, const double* grn, int dim, const double* C, const double* delta6,
                      const double* Areg, double* Gs, double* gb, double* gC, double* gr) {
    step_backward(gCn, grn, dim, C, delta6, Areg, Gs, gb, gC, gr);
}
double hc_kabsch_forward(const double* acc, double* C, double* r, double* save) { return kabsch_forward(acc, C, r, save); }
void hc_kabsch_backward(const double* gC, const double* gr, const double* save, double* gacc) { kabsch_backward(gC, gr, save, gacc); }
void hc_svd3(const double* A, double* U, double* S, double* V) { svd3(A, U, S, V); }
int hc_sizeof_params() { return (int)sizeof(WeightParams); }

}  // extern "C"
