// Per-point and per-cloud arithmetic of the differentiable-ICP iteration.
//
// Everything here is plain inline C++ templated on the scalar type, included by
// the HIP kernels (dicp_kernels.hip) — and by tests/hostcheck/hostcheck.cpp, a
// TEST-ONLY g++ build that checks these formulas against autograd on a CPU box
// with no GPU.  The product never runs the host instantiation.
//
// Reference semantics (file:line under /root/reference/dICP):
//   transform            ICP.py:137          residuals  ICP.py:143-149
//   trim weight          loss.py:43-58       huber      loss.py:21-32   cauchy loss.py:34-41
//   weight combine       ICP.py:162-169,194-196
//   Jacobian             ICP.py:171-190,513-531
//   normal equations     ICP.py:198-201      pose update ICP.py:209-217
// Backward: closed-form adjoint of the above (the reference uses stock autograd).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define DICP_HD __host__ __device__ __forceinline__
#else
#define DICP_HD inline
#endif

namespace dicp {

enum Mode { MODE_PT2PT = 0, MODE_PT2PL = 1 };
enum Loss { LOSS_NONE = 0, LOSS_HUBER = 1, LOSS_CAUCHY = 2, LOSS_TRIM = 3 };   // trim as loss_fn: loss.py:15-16

// Accumulator slots produced per cloud and iteration.
//   [0,21)  upper triangle of the 6x6  A = sum u J^T J   (row-major, i<=j)
//   [21,27) b = sum u J^T e
//   27 cost = sum u e^2      28 sum of w      29 #(w > match_thresh) (x3 for pt2pt rows)
constexpr int ACC_A = 0, ACC_B = 21, ACC_COST = 27, ACC_SUMW = 28, ACC_NMATCH = 29, NACC = 30;
constexpr int NACC_PAD = 32;
// Backward per-cloud slots: C-bar (9, row-major) then r-bar (3).
constexpr int NBWD = 12, NBWD_PAD = 16;

struct WeightParams {
    int mode;          // Mode
    int trim_on;       // trim_dist is not None and >= 0          (ICP.py:153)
    int differentiable;
    int loss;          // Loss
    double trim_dist;  // tau
    double tanh_k;     // tanh_steepness
    double loss_delta; // loss_fn["metric"]
    double match_thresh;
};

// A parameter of the call in the kernel's scalar type.  On the device a float comes back in a SCALAR register: the parameters are doubles, a conversion is a
// vector instruction, and its wave-uniform result would otherwise sit in a vector register for as long as a loop around the point functions runs.
template <typename T> DICP_HD T wp_val(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (sizeof(T) == 4) return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)v)));
#endif
    return (T)v;
}

DICP_HD int tri(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }   // i <= j

// float32 on the device: the hardware's own one-instruction forms (v_sqrt_f32, v_rcp_f32, v_exp_f32: 1 ulp each), not the correctly rounded expansions --
// a division is 10 vector instructions that way, tanhf ~40, sqrtf ~12, and the per-point functions below are instruction-bound: of the 430 vector instructions
// the windowed backward spent per point, ~100 were these (profiles/r05_point_math.txt).  A weight or a gradient moves by parts in 1e7; north_star's float32
// bar is 1e-4 / 1e-3, and every form of every kernel shares these functions, so results that are compared bit for bit (searches, certified iterations) still are.
// float64, and the host build of this header (tests/hostcheck), keep the correctly rounded forms.
// Edges of the one-instruction forms (they differ from IEEE division beyond the last ulp): v_rcp_f32 flushes denormals -- m_div(a, b) is +-inf for |b| < 2^-126
// and 0 for |b| > 2^126 -- and a * rcp(a) is not exactly 1: a Huber / Cauchy weight at a residual of exactly the metric comes out 0.99999994.  The quotients here
// are metric / residual, 1 / (1 + (e/c)^2) and trim / loss ratios of coordinates in metres: neither range occurs; 0 / 0 and x / 0 stay NaN / inf as in the
// reference (the hard Huber slope's NaN at a zero residual is kept: tests/test_gpu_parity.py).
#if defined(__HIP_DEVICE_COMPILE__)
DICP_HD float  m_sqrt(float x)  { return __builtin_amdgcn_sqrtf(x); }
DICP_HD float  m_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
DICP_HD float  m_tanh(float x)  {                       // 1 - 2 / (e^2x + 1): +-1 at the ends (e -> inf, 0), absolute error ~1e-7
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
#else
DICP_HD float  m_sqrt(float x)  { return sqrtf(x); }
DICP_HD float  m_div(float a, float b) { return a / b; }
DICP_HD float  m_tanh(float x)  { return tanhf(x); }
#endif
DICP_HD double m_sqrt(double x) { return sqrt(x); }
DICP_HD double m_div(double a, double b) { return a / b; }
DICP_HD double m_tanh(double x) { return tanh(x); }
DICP_HD float  m_abs(float x)   { return fabsf(x); }
DICP_HD double m_abs(double x)  { return fabs(x); }

// d/d(en) of the hard Huber weight where(en > delta, delta/en, 1), loss.py:32.  autograd differentiates
// BOTH where() branches and masks afterwards, so at en == 0 the reference's gradient is 0 * (-inf) = NaN;
// that (tested-as-is) behaviour is kept rather than silently repaired.
template <typename T> DICP_HD T hard_huber_slope(T en, T delta) {
    if (en > delta) return m_div(-delta, en * en);
    return (en == T(0)) ? (T(0) * m_div(-delta, en * en)) : T(0);
}

template <typename T> DICP_HD void cross3(const T* a, const T* b, T* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename T> DICP_HD T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename T> DICP_HD void matvec3(const T* C, const T* p, T* o) {
    o[0] = C[0] * p[0] + C[1] * p[1] + C[2] * p[2];
    o[1] = C[3] * p[0] + C[4] * p[1] + C[5] * p[2];
    o[2] = C[6] * p[0] + C[7] * p[1] + C[8] * p[2];
}

// ---------------------------------------------------------------- per-point state
template <typename T> struct PointState {
    T q[3];      // C p
    T e3[3];     // C p + r - y
    T d3;        // ||e3||
    T e;         // pt2pl: n . e3
    T en;        // norm the robust loss sees (|e| or d3)
    T tw, lw;    // trim and loss weights
    T th;        // tanh(...) of the soft trim gate
    T lth;       // tanh(...) of the soft gate when loss_fn itself is "trim" (loss.py:15-16)
    T w, root, ws, u;   // w, sqrt(w+1e-10), root-1e-5, ws^2
};

template <typename T, int MODE>
DICP_HD void point_weights(const WeightParams& P, const T* C, const T* r, const T* p, const T* y,
                           const T* nrm, T w0, PointState<T>& s) {
    matvec3(C, p, s.q);
    s.e3[0] = s.q[0] + r[0] - y[0];
    s.e3[1] = s.q[1] + r[1] - y[1];
    s.e3[2] = s.q[2] + r[2] - y[2];
    s.d3 = m_sqrt(dot3(s.e3, s.e3));
    if (MODE == MODE_PT2PL) { s.e = dot3(s.e3, nrm); s.en = m_abs(s.e); }
    else                    { s.e = T(0);            s.en = s.d3; }
    s.tw = T(1); s.th = T(0);
    if (P.trim_on) {
        if (P.differentiable) {                                            // loss.py:54
            s.th = m_tanh(wp_val<T>(P.tanh_k) * (wp_val<T>(P.trim_dist) - s.d3) - T(3));
            s.tw = T(0.5) * s.th + T(0.5);
        } else {                                                           // loss.py:58
            s.tw = (s.d3 < wp_val<T>(P.trim_dist)) ? T(1) : T(0);
        }
    }
    s.lw = T(1); s.lth = T(0);
    const T dl = wp_val<T>(P.loss_delta);
    if (P.loss == LOSS_HUBER) {
        if (P.differentiable) s.lw = m_div(dl * dl, dl * dl + s.en * s.en);  // loss.py:30
        else                  s.lw = (s.en > dl) ? m_div(dl, s.en) : T(1);   // loss.py:32
    } else if (P.loss == LOSS_CAUCHY) {                                    // loss.py:41
        const T t = m_div(s.en, dl);
        s.lw = m_div(T(1), T(1) + t * t);
    } else if (P.loss == LOSS_TRIM) {                                      // loss.py:43-58 on the loss residual (ICP.py:157-160)
        if (P.differentiable) {
            s.lth = m_tanh(wp_val<T>(P.tanh_k) * (dl - s.en) - T(3));
            s.lw = T(0.5) * s.lth + T(0.5);
        } else {
            s.lw = (s.en < dl) ? T(1) : T(0);
        }
    }
    s.w = w0 * s.tw * s.lw;                                                // ICP.py:169
    s.root = m_sqrt(s.w + T(1.0e-10));                                     // ICP.py:194
    s.ws = s.root - T(1.0e-5);
    s.u = s.ws * s.ws;
}

// Forward: add this point's contribution to acc[NACC]; returns w through s.w.
template <typename T, int MODE>
DICP_HD void point_forward(const WeightParams& P, const T* C, const T* r, const T* p, const T* y,
                           const T* nrm, T w0, T* acc, PointState<T>& s) {
    point_weights<T, MODE>(P, C, r, p, y, nrm, w0, s);
    const T u = s.u;
    if (MODE == MODE_PT2PL) {
        T j[6];
        cross3(nrm, s.q, j);                       // (q^)^T n = n x q        ICP.py:175
        j[3] = -nrm[0]; j[4] = -nrm[1]; j[5] = -nrm[2];                    // ICP.py:176
        int k = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            const T uj = u * j[a];
#pragma unroll
            for (int b = a; b < 6; ++b) acc[ACC_A + k++] += uj * j[b];
            acc[ACC_B + a] += uj * s.e;
        }
        acc[ACC_COST] += u * s.e * s.e;
    } else {
        const T* q = s.q;
        const T* e = s.e3;
        const T qq = dot3(q, q);
        // J = [q^, -I]: J^T J = [[|q|^2 I - q q^T, q^], [-q^, I]]        ICP.py:178-183
        acc[ACC_A + 0]  += u * (qq - q[0] * q[0]);   // (0,0)
        acc[ACC_A + 1]  += u * (-q[0] * q[1]);       // (0,1)
        acc[ACC_A + 2]  += u * (-q[0] * q[2]);       // (0,2)
        /* (0,3) = 0 */
        acc[ACC_A + 4]  += u * (-q[2]);              // (0,4)
        acc[ACC_A + 5]  += u * (q[1]);               // (0,5)
        acc[ACC_A + 6]  += u * (qq - q[1] * q[1]);   // (1,1)
        acc[ACC_A + 7]  += u * (-q[1] * q[2]);       // (1,2)
        acc[ACC_A + 8]  += u * (q[2]);               // (1,3)
        /* (1,4) = 0 */
        acc[ACC_A + 10] += u * (-q[0]);              // (1,5)
        acc[ACC_A + 11] += u * (qq - q[2] * q[2]);   // (2,2)
        acc[ACC_A + 12] += u * (-q[1]);              // (2,3)
        acc[ACC_A + 13] += u * (q[0]);               // (2,4)
        /* (2,5) = 0 */
        acc[ACC_A + 15] += u;                        // (3,3)
        acc[ACC_A + 18] += u;                        // (4,4)
        acc[ACC_A + 20] += u;                        // (5,5)
        T exq[3];
        cross3(e, q, exq);                           // (q^)^T e = e x q
        acc[ACC_B + 0] += u * exq[0];
        acc[ACC_B + 1] += u * exq[1];
        acc[ACC_B + 2] += u * exq[2];
        acc[ACC_B + 3] -= u * e[0];
        acc[ACC_B + 4] -= u * e[1];
        acc[ACC_B + 5] -= u * e[2];
        acc[ACC_COST] += u * dot3(e, e);
    }
    const T rows = (MODE == MODE_PT2PT) ? T(3) : T(1);
    acc[ACC_SUMW] += rows * s.w;
    if (s.w > wp_val<T>(P.match_thresh)) acc[ACC_NMATCH] += rows;
}

// Backward for one point.  Gs = G_A + G_A^T (6x6 row-major, symmetric), gb = dL/db.
// Outputs: gp (dL/dp), gy (dL/dy), gn (dL/dnormal, pt2pl only), gw0 (dL/dw0),
// and adds q-bar p^T into gC[9] and s-bar into gr[3].
template <typename T, int MODE>
DICP_HD void point_backward(const WeightParams& P, const T* C, const T* r, const T* p, const T* y,
                            const T* nrm, T w0, const T* Gs, const T* gb,
                            T* gp, T* gy, T* gn, T& gw0, T* gC, T* gr) {
    PointState<T> s;
    point_weights<T, MODE>(P, C, r, p, y, nrm, w0, s);
    const T u = s.u;
    T ubar, qbar[3], e3bar[3] = {T(0), T(0), T(0)};
    T ebar_s = T(0);            // pt2pl scalar e-bar
    T ja[3] = {T(0), T(0), T(0)}, jc[3] = {T(0), T(0), T(0)};   // pt2pl J-bar halves
    T j[6];
    if (MODE == MODE_PT2PL) {
        cross3(nrm, s.q, j);
        j[3] = -nrm[0]; j[4] = -nrm[1]; j[5] = -nrm[2];
        T Gj[6];
        T jGj = T(0), jgb = T(0);
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            T acc = T(0);
#pragma unroll
            for (int b = 0; b < 6; ++b) acc += Gs[a * 6 + b] * j[b];
            Gj[a] = acc;
            jGj += j[a] * acc;
            jgb += j[a] * gb[a];
        }
        ubar = T(0.5) * jGj + s.e * jgb;
        ebar_s = u * jgb;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ja[a] = u * (Gj[a] + s.e * gb[a]);
            jc[a] = u * (Gj[a + 3] + s.e * gb[a + 3]);
        }
    } else {
        const T* q = s.q;
        const T* e = s.e3;
        // Pm = Q G11 - G21, Rm = Q G12 - G22 with Q = q^ ; column c of Q G = q x G[:,c]
        T Pm[9], Rm[9];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            T g1[3] = {Gs[0 * 6 + c], Gs[1 * 6 + c], Gs[2 * 6 + c]};
            T g2[3] = {Gs[0 * 6 + 3 + c], Gs[1 * 6 + 3 + c], Gs[2 * 6 + 3 + c]};
            T x1[3], x2[3];
            cross3(q, g1, x1);
            cross3(q, g2, x2);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                Pm[i * 3 + c] = x1[i] - Gs[(3 + i) * 6 + c];
                Rm[i * 3 + c] = x2[i] - Gs[(3 + i) * 6 + 3 + c];
            }
        }
        // rows of Q
        const T Q[9] = {T(0), -q[2], q[1], q[2], T(0), -q[0], -q[1], q[0], T(0)};
        T tr = T(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int c = 0; c < 3; ++c) tr += Pm[i * 3 + c] * Q[i * 3 + c];
            tr -= Rm[i * 3 + i];
        }
        T Jgb[3], qxg[3];
        cross3(q, gb, qxg);
        Jgb[0] = qxg[0] - gb[3]; Jgb[1] = qxg[1] - gb[4]; Jgb[2] = qxg[2] - gb[5];
        ubar = T(0.5) * tr + dot3(e, Jgb);
        // M = u (Pm + e gb1^T); q-bar from the skew structure of Q
        T M[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) M[i * 3 + c] = u * (Pm[i * 3 + c] + e[i] * gb[c]);
        qbar[0] = M[7] - M[5];
        qbar[1] = M[2] - M[6];
        qbar[2] = M[3] - M[1];
        e3bar[0] = u * Jgb[0]; e3bar[1] = u * Jgb[1]; e3bar[2] = u * Jgb[2];
    }
    // u = (sqrt(w+1e-10) - 1e-5)^2
    const T wbar = m_div(ubar * s.ws, s.root);
    gw0 = wbar * s.tw * s.lw;
    const T twbar = wbar * w0 * s.lw;
    const T lwbar = wbar * w0 * s.tw;
    // robust loss -> error
    const T dl = wp_val<T>(P.loss_delta);
    T dlw_den = T(0);     // d lw / d en
    if (P.loss == LOSS_HUBER) {
        if (P.differentiable) dlw_den = m_div(-T(2) * s.en * s.lw * s.lw, dl * dl);
        else                  dlw_den = hard_huber_slope(s.en, dl);
    } else if (P.loss == LOSS_CAUCHY) {
        dlw_den = m_div(-T(2) * s.en * s.lw * s.lw, dl * dl);
    } else if (P.loss == LOSS_TRIM && P.differentiable) {                  // the hard gate has no gradient
        dlw_den = -T(0.5) * wp_val<T>(P.tanh_k) * (T(1) - s.lth * s.lth);
    }
    if (MODE == MODE_PT2PL) {
        // en = |e| ; torch's norm backward gives e/|e| (0 at e == 0)
        const T sgn = (s.e > T(0)) ? T(1) : ((s.e < T(0)) ? T(-1) : T(0));
        ebar_s += lwbar * dlw_den * sgn;
    } else {
        // torch's vector-norm backward is e3/|e3|, and 0 at e3 == 0 (a NaN slope still propagates)
        const T f = (s.d3 > T(0)) ? m_div(lwbar * dlw_den, s.d3) : lwbar * dlw_den * T(0);
        e3bar[0] += f * s.e3[0]; e3bar[1] += f * s.e3[1]; e3bar[2] += f * s.e3[2];
    }
    // soft trim gate -> e3 (hard gate has no gradient)
    if (P.trim_on && P.differentiable && s.d3 > T(0)) {
        const T f = m_div(twbar * (-T(0.5) * wp_val<T>(P.tanh_k) * (T(1) - s.th * s.th)), s.d3);
        e3bar[0] += f * s.e3[0]; e3bar[1] += f * s.e3[1]; e3bar[2] += f * s.e3[2];
    }
    if (MODE == MODE_PT2PL) {
        e3bar[0] += ebar_s * nrm[0]; e3bar[1] += ebar_s * nrm[1]; e3bar[2] += ebar_s * nrm[2];
        T qxa[3];
        cross3(s.q, ja, qxa);
        gn[0] = ebar_s * s.e3[0] + qxa[0] - jc[0];
        gn[1] = ebar_s * s.e3[1] + qxa[1] - jc[1];
        gn[2] = ebar_s * s.e3[2] + qxa[2] - jc[2];
        cross3(ja, nrm, qbar);
    } else {
        gn[0] = gn[1] = gn[2] = T(0);
    }
    gy[0] = -e3bar[0]; gy[1] = -e3bar[1]; gy[2] = -e3bar[2];
    qbar[0] += e3bar[0]; qbar[1] += e3bar[1]; qbar[2] += e3bar[2];
    gr[0] += e3bar[0]; gr[1] += e3bar[1]; gr[2] += e3bar[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) gC[i * 3 + c] += qbar[i] * p[c];
    gp[0] = C[0] * qbar[0] + C[3] * qbar[1] + C[6] * qbar[2];
    gp[1] = C[1] * qbar[0] + C[4] * qbar[1] + C[7] * qbar[2];
    gp[2] = C[2] * qbar[0] + C[5] * qbar[1] + C[8] * qbar[2];
}

// ------------------------------------------------------------------ per-cloud step
// All in double regardless of the cloud dtype (36 numbers per cloud).

// Rodrigues coefficients: exp(phi^) = I + a K + b K^2, J_l = a I + c phi phi^T + b K.
DICP_HD void so3_coeffs(const double* phi, double& a, double& b, double& c) {
    const double t2 = phi[0] * phi[0] + phi[1] * phi[1] + phi[2] * phi[2];
    if (t2 < 1e-8) {
        a = 1.0 - t2 / 6.0 + t2 * t2 / 120.0;
        b = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
        c = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        const double t = sqrt(t2);
        a = sin(t) / t;
        b = (1.0 - cos(t)) / t2;
        c = (1.0 - a) / t2;
    }
}

DICP_HD void so3_exp(const double* phi, double* R) {
    double a, b, c;
    so3_coeffs(phi, a, b, c);
    const double x = phi[0], y = phi[1], z = phi[2];
    // K^2 = phi phi^T - |phi|^2 I
    const double t2 = x * x + y * y + z * z;
    R[0] = 1.0 + b * (x * x - t2); R[1] = -a * z + b * x * y;     R[2] = a * y + b * x * z;
    R[3] = a * z + b * x * y;      R[4] = 1.0 + b * (y * y - t2); R[5] = -a * x + b * y * z;
    R[6] = -a * y + b * x * z;     R[7] = a * x + b * y * z;      R[8] = 1.0 + b * (z * z - t2);
}

// Expand the 21-slot upper triangle into a full symmetric 6x6.
template <typename T> DICP_HD void unpack_sym6(const T* tri21, double* A) {
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) { A[i * 6 + j] = A[j * 6 + i] = (double)tri21[k++]; }
}

// Map between the d-dim unknowns and the 6-vector: dim==2 optimises (rot-z, tx, ty) = slots 2,3,4
// (ICP.py:186-189, 203-207).
DICP_HD int slot(int dim, int i) { return dim == 2 ? i + 2 : i; }
DICP_HD int ndof(int dim) { return dim == 2 ? 3 : 6; }

// Forward step.  A6 (full 6x6, without the regulariser), b6 -> delta6, and the new pose.
// Areg receives the d x d matrix actually inverted (leading dim 6), kept for backward.
// D x D solve with partial pivoting (what LAPACK getrf does under torch.linalg.inv, ICP.py:201), fully unrolled with compile-time indices so that the matrix stays in
// REGISTERS on the GPU (a runtime-indexed private array would live in scratch memory, and an LDS copy makes the
// serial elimination LDS-latency-bound: the per-cloud step kernels sit between the big launches, so their
// latency is what small problems see).  Returns false if a pivot is exactly zero.
template <int D>
DICP_HD bool solve_fixed(double (&M)[D * D], double (&rhs)[D], double (&x)[D]) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
        int piv = k;
        double best = fabs(M[k * D + k]);
#pragma unroll
        for (int i = k + 1; i < D; ++i) {
            const double v = fabs(M[i * D + k]);
            if (v > best) { best = v; piv = i; }
        }
        if (best == 0.0) return false;
#pragma unroll
        for (int i = k + 1; i < D; ++i) {               // branch-free row swap k <-> piv
            const bool sw = (piv == i);
#pragma unroll
            for (int c = k; c < D; ++c) {
                const double a = M[k * D + c], b = M[i * D + c];
                M[k * D + c] = sw ? b : a;
                M[i * D + c] = sw ? a : b;
            }
            const double a = rhs[k], b = rhs[i];
            rhs[k] = sw ? b : a;
            rhs[i] = sw ? a : b;
        }
        const double inv = 1.0 / M[k * D + k];
#pragma unroll
        for (int i = k + 1; i < D; ++i) {
            const double f = M[i * D + k] * inv;
#pragma unroll
            for (int c = k + 1; c < D; ++c) M[i * D + c] -= f * M[k * D + c];
            rhs[i] -= f * rhs[k];
        }
    }
#pragma unroll
    for (int i = D - 1; i >= 0; --i) {
        double v = rhs[i];
#pragma unroll
        for (int c = i + 1; c < D; ++c) v -= M[i * D + c] * x[c];
        x[i] = v / M[i * D + i];
    }
    return true;
}

// the solve alone: delta6 (zeros outside the optimised slots) and the matrix that was inverted
template <int D>
DICP_HD void step_solve_fixed(const double* A6, const double* b6, double* delta6, double* Areg) {
    constexpr int OFF = (D == 3) ? 2 : 0;              // dim == 2 optimises slots 2,3,4 (ICP.py:186-189)
    double M[D * D], rhs[D], x[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) M[i * D + j] = A6[(i + OFF) * 6 + (j + OFF)];
        M[i * D + i] += 1e-12;                                      // ICP.py:200
        rhs[i] = b6[i + OFF];
    }
    for (int i = 0; i < 36; ++i) Areg[i] = 0.0;
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
        for (int j = 0; j < D; ++j) Areg[i * 6 + j] = M[i * D + j];
    for (int i = 0; i < 6; ++i) delta6[i] = 0.0;
    if (solve_fixed<D>(M, rhs, x)) {
#pragma unroll
        for (int i = 0; i < D; ++i) delta6[i + OFF] = -x[i];         // ICP.py:201
    }
}

DICP_HD void step_solve(const double* A6, const double* b6, int dim, double* delta6, double* Areg) {
    if (dim == 2) step_solve_fixed<3>(A6, b6, delta6, Areg);
    else          step_solve_fixed<6>(A6, b6, delta6, Areg);
}

// solve + the new pose from the UNROUNDED step (the kernels round delta to the cloud type first, like the reference, and apply it themselves: step_body)
DICP_HD void step_forward(const double* A6, const double* b6, int dim, const double* C, const double* r,
                          double* delta6, double* Cn, double* rn, double* Areg) {
    step_solve(A6, b6, dim, delta6, Areg);
    double R[9];
    so3_exp(delta6, R);                                              // ICP.py:210
    for (int i = 0; i < 3; ++i)                                      // C <- R^T C   ICP.py:214
        for (int j = 0; j < 3; ++j)
            Cn[i * 3 + j] = R[0 * 3 + i] * C[0 * 3 + j] + R[1 * 3 + i] * C[1 * 3 + j] + R[2 * 3 + i] * C[2 * 3 + j];
    for (int i = 0; i < 3; ++i) rn[i] = r[i] - delta6[3 + i];        // ICP.py:216
}

// Backward step.  In: gCn, grn (cotangents of the new pose), saved C, delta6, Areg.
// Out: Gs = G_A + G_A^T (6x6), gb (6), gC, gr (cotangents of the old pose).
template <int D>
DICP_HD void step_backward_fixed(const double* gCn, const double* grn, const double* C,
                                 const double* delta6, const double* Areg,
                                 double* Gs, double* gb, double* gC, double* gr) {
    constexpr int OFF = (D == 3) ? 2 : 0;
    double R[9];
    so3_exp(delta6, R);
    // C_new = R^T C  ->  gC = R gCn ,  gR = C gCn^T
    double gR[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            gC[i * 3 + j] = R[i * 3 + 0] * gCn[0 * 3 + j] + R[i * 3 + 1] * gCn[1 * 3 + j] + R[i * 3 + 2] * gCn[2 * 3 + j];
            gR[i * 3 + j] = C[i * 3 + 0] * gCn[j * 3 + 0] + C[i * 3 + 1] * gCn[j * 3 + 1] + C[i * 3 + 2] * gCn[j * 3 + 2];
        }
    for (int i = 0; i < 3; ++i) gr[i] = grn[i];
    // dR = (J_l dphi)^ R  ->  gphi = J_l^T vee(M - M^T), M = gR R^T
    double Mm[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Mm[i * 3 + j] = gR[i * 3 + 0] * R[j * 3 + 0] + gR[i * 3 + 1] * R[j * 3 + 1] + gR[i * 3 + 2] * R[j * 3 + 2];
    const double v[3] = {Mm[7] - Mm[5], Mm[2] - Mm[6], Mm[3] - Mm[1]};
    double a, b, c;
    so3_coeffs(delta6, a, b, c);
    const double* phi = delta6;
    const double pv = phi[0] * v[0] + phi[1] * v[1] + phi[2] * v[2];
    double pxv[3];
    cross3(phi, v, pxv);
    double gd[6];
    for (int i = 0; i < 3; ++i) gd[i] = a * v[i] + c * phi[i] * pv - b * pxv[i];
    for (int i = 0; i < 3; ++i) gd[3 + i] = -grn[i];
    // delta = -Areg^{-1} b  ->  g = Areg^{-1} gdelta ; gb = -g ; G_A = -g delta^T
    double Mx[D * D], rhs[D], g[D];
#pragma unroll
    for (int i = 0; i < D; ++i) {
#pragma unroll
        for (int j = 0; j < D; ++j) Mx[i * D + j] = Areg[i * 6 + j];
        rhs[i] = gd[i + OFF];
    }
    for (int i = 0; i < 36; ++i) Gs[i] = 0.0;
    for (int i = 0; i < 6; ++i) gb[i] = 0.0;
    if (!solve_fixed<D>(Mx, rhs, g)) return;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        gb[i + OFF] = -g[i];
#pragma unroll
        for (int j = 0; j < D; ++j) Gs[(i + OFF) * 6 + (j + OFF)] = -(g[i] * delta6[j + OFF] + delta6[i + OFF] * g[j]);
    }
}

DICP_HD void step_backward(const double* gCn, const double* grn, int dim, const double* C,
                           const double* delta6, const double* Areg,
                           double* Gs, double* gb, double* gC, double* gr) {
    if (dim == 2) step_backward_fixed<3>(gCn, grn, C, delta6, Areg, Gs, gb, gC, gr);
    else          step_backward_fixed<6>(gCn, grn, C, delta6, Areg, Gs, gb, gC, gr);
}

// ------------------------------------------------------------------ Kabsch / SVD step
// Closed-form point-to-point alignment (the step of the reference's pt2pt_dICP_SVD, ICP.py:557-573,
// with the rotation composed as U diag(1,1,det U det V) V^T -- the reference multiplies by V where V^T is
// required, which is only right for planar data).  Batched and weighted:
//   sums per cloud: S0 = sum w, sp = sum w p, sy = sum w y, M = sum w y p^T, pp = sum w |p|^2, yy = sum w |y|^2
//   W = M/S0 - mu_t mu_s^T ;  W = U S V^T ;  C = U D V^T ;  r = mu_t - C mu_s
constexpr int KAB_S0 = 0, KAB_SP = 1, KAB_SY = 4, KAB_M = 7, KAB_PP = 16, KAB_YY = 17, NKAB = 18;
constexpr int KAB_SAVE = 40;      // doubles kept per cloud for the backward pass

// One-sided Jacobi SVD of a 3x3 (row-major): A = U diag(S) V^T, S descending, U and V orthogonal.
// A pair of columns counts as orthogonal at 1e-15 of the product of their norms: a few units of double rounding.  (Up to round 3 the bound was 1e-17 --
// below the rounding of the inner product itself, so it was never met and EVERY call ran all 30 sweeps of three rotations, each with three square roots
// and two divisions in double: 45 of the step kernel's 48 us.  Jacobi converges quadratically; 4-6 sweeps reach 1e-15.)
// Every index below is a compile-time constant (the pairs and the column sort are written out): with loops over (p, q) and an index array for the
// sort the nine + nine entries lived in scratch memory and one lane's 3x3 SVD took 45 us of the SVD loop's step kernel (profiles/r04_svd_step.txt).
#define DICP_SVD3_ROTATE(p, q)                                                                                              \
    {                                                                                                                       \
        double a = 0, b = 0, g = 0;                                                                                         \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) { a += G[i * 3 + p] * G[i * 3 + p]; b += G[i * 3 + q] * G[i * 3 + q]; g += G[i * 3 + p] * G[i * 3 + q]; } \
        if (!(fabs(g) <= 1e-300 || fabs(g) <= 1e-15 * sqrt(a * b))) {                                                       \
            off += fabs(g);                                                                                                 \
            const double zeta = (b - a) / (2.0 * g);                                                                        \
            const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));                             \
            const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;                                                           \
            _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                                 \
                const double gp = G[i * 3 + p], gq = G[i * 3 + q];                                                          \
                G[i * 3 + p] = c * gp - sn * gq; G[i * 3 + q] = sn * gp + c * gq;                                           \
                const double vp = V[i * 3 + p], vq = V[i * 3 + q];                                                          \
                V[i * 3 + p] = c * vp - sn * vq; V[i * 3 + q] = sn * vp + c * vq;                                           \
            }                                                                                                               \
        }                                                                                                                   \
    }
// columns a and b of G and V change places when column b's norm is the larger one (strictly: the order of equal values stays)
#define DICP_SVD3_ORDER(a, b)                                                                                               \
    if (nrm[b] > nrm[a]) {                                                                                                  \
        const double tn = nrm[a]; nrm[a] = nrm[b]; nrm[b] = tn;                                                             \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                                     \
            const double tg = G[i * 3 + a]; G[i * 3 + a] = G[i * 3 + b]; G[i * 3 + b] = tg;                                 \
            const double tv = V[i * 3 + a]; V[i * 3 + a] = V[i * 3 + b]; V[i * 3 + b] = tv;                                 \
        }                                                                                                                   \
    }
DICP_HD void svd3(const double* A, double* U, double* S, double* V) {
    double G[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { G[i] = A[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
#pragma unroll 1
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        DICP_SVD3_ROTATE(0, 1)
        DICP_SVD3_ROTATE(0, 2)
        DICP_SVD3_ROTATE(1, 2)
        if (off == 0.0) break;
    }
    double nrm[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) nrm[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
    DICP_SVD3_ORDER(0, 1)                                              // sort columns by singular value, descending
    DICP_SVD3_ORDER(0, 2)
    DICP_SVD3_ORDER(1, 2)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        S[j] = nrm[j];
#pragma unroll
        for (int i = 0; i < 3; ++i) U[i * 3 + j] = (S[j] > 0) ? G[i * 3 + j] / S[j] : 0.0;
    }
    // rank-deficient input (planar / collinear clouds): complete U to an orthonormal basis
    const double tiny = 1e-14 * (S[0] > 0 ? S[0] : 1.0);
    if (S[0] <= 0) { for (int i = 0; i < 9; ++i) U[i] = (i % 4 == 0) ? 1.0 : 0.0; return; }
    if (S[1] <= tiny) {
        const double u0[3] = {U[0], U[3], U[6]};
        int k = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0 : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
        const double e[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
        const double d = k == 0 ? u0[0] : (k == 1 ? u0[1] : u0[2]);
        double v[3] = {e[0] - d * u0[0], e[1] - d * u0[1], e[2] - d * u0[2]};
        const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        U[1] = v[0] / n; U[4] = v[1] / n; U[7] = v[2] / n;
    }
    if (S[2] <= tiny) {
        const double a0[3] = {U[0], U[3], U[6]}, a1[3] = {U[1], U[4], U[7]};
        double c3[3];
        cross3(a0, a1, c3);
        U[2] = c3[0]; U[5] = c3[1]; U[8] = c3[2];
    }
}

DICP_HD double det3(const double* A) {
    return A[0] * (A[4] * A[8] - A[5] * A[7]) - A[1] * (A[3] * A[8] - A[5] * A[6]) + A[2] * (A[3] * A[7] - A[4] * A[6]);
}

// sums -> pose.  save[KAB_SAVE]: U(9) V(9) lambda(3) d3(1) mus(3) mut(3) S0(1) C(9) = 38 used.
// Returns the alignment cost sum w |C p + r - y|^2 under the NEW pose (ICP.py:585's stopping quantity).
DICP_HD double kabsch_forward(const double* acc, double* C, double* r, double* save) {
    const double S0 = acc[KAB_S0];
    for (int i = 0; i < 9; ++i) C[i] = (i % 4 == 0) ? 1.0 : 0.0;
    r[0] = r[1] = r[2] = 0.0;
    for (int i = 0; i < KAB_SAVE; ++i) save[i] = 0.0;
    if (!(S0 > 0.0)) return 0.0;                       // no weight at all (empty / switched-off cloud): identity
    double mus[3], mut[3], W[9];
    for (int a = 0; a < 3; ++a) { mus[a] = acc[KAB_SP + a] / S0; mut[a] = acc[KAB_SY + a] / S0; }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) W[a * 3 + b] = acc[KAB_M + a * 3 + b] / S0 - mut[a] * mus[b];
    double U[9], S[3], V[9];
    svd3(W, U, S, V);
    const double d3 = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;          // ICP.py:570
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) C[a * 3 + b] = U[a * 3 + 0] * V[b * 3 + 0] + U[a * 3 + 1] * V[b * 3 + 1] + d3 * U[a * 3 + 2] * V[b * 3 + 2];
    for (int a = 0; a < 3; ++a) r[a] = mut[a] - (C[a * 3] * mus[0] + C[a * 3 + 1] * mus[1] + C[a * 3 + 2] * mus[2]);   // ICP.py:573
    for (int i = 0; i < 9; ++i) { save[i] = U[i]; save[9 + i] = V[i]; save[29 + i] = C[i]; }
    save[18] = S[0]; save[19] = S[1]; save[20] = d3 * S[2]; save[21] = d3;
    for (int a = 0; a < 3; ++a) { save[22 + a] = mus[a]; save[25 + a] = mut[a]; }
    save[28] = S0;
    // sum w |Cp + r - y|^2 = pp + yy + S0|r|^2 + 2 r.C sp - 2 r.sy - 2 <C, M>
    double rr = 0, rCsp = 0, rsy = 0, CM = 0;
    for (int a = 0; a < 3; ++a) {
        rr += r[a] * r[a];
        rsy += r[a] * acc[KAB_SY + a];
        for (int b = 0; b < 3; ++b) { rCsp += r[a] * C[a * 3 + b] * acc[KAB_SP + b]; CM += C[a * 3 + b] * acc[KAB_M + a * 3 + b]; }
    }
    return acc[KAB_PP] + acc[KAB_YY] + S0 * rr + 2.0 * rCsp - 2.0 * rsy - 2.0 * CM;
}

// Adjoint of kabsch_forward: cotangents of (C, r) -> cotangents of the sums [gS0, gsp(3), gsy(3), gM(9)].
// Rotation: with P = C^T W = V Lambda V^T (Lambda = D S) and dC = C Om, Om skew:  Om~_ij =
// (d_i X_ij - d_j X_ji) / (lambda_i + lambda_j), X = U^T dW V.  Degenerate pairs (lambda_i + lambda_j ~ 0) get 0.
DICP_HD void kabsch_backward(const double* gC, const double* gr, const double* save, double* gacc) {
    for (int i = 0; i < 16; ++i) gacc[i] = 0.0;
    const double S0 = save[28];
    if (!(S0 > 0.0)) return;
    const double* U = save; const double* V = save + 9; const double* lam = save + 18; const double* C = save + 29;
    const double d[3] = {1.0, 1.0, save[21]};
    const double* mus = save + 22; const double* mut = save + 25;
    // r = mut - C mus
    double gmut[3], gmus[3], gCt[9];
    for (int a = 0; a < 3; ++a) {
        gmut[a] = gr[a];
        gmus[a] = -(C[0 * 3 + a] * gr[0] + C[1 * 3 + a] * gr[1] + C[2 * 3 + a] * gr[2]);
        for (int b = 0; b < 3; ++b) gCt[a * 3 + b] = gC[a * 3 + b] - gr[a] * mus[b];
    }
    // G = D U^T gCt V
    double T1[9], G[9], Xb[9], gW[9];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) T1[a * 3 + b] = U[0 * 3 + a] * gCt[0 * 3 + b] + U[1 * 3 + a] * gCt[1 * 3 + b] + U[2 * 3 + a] * gCt[2 * 3 + b];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) G[a * 3 + b] = d[a] * (T1[a * 3 + 0] * V[0 * 3 + b] + T1[a * 3 + 1] * V[1 * 3 + b] + T1[a * 3 + 2] * V[2 * 3 + b]);
    const double scale = fabs(lam[0]) + fabs(lam[1]) + fabs(lam[2]);
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const double den = lam[a] + lam[b];
            Xb[a * 3 + b] = (a != b && fabs(den) > 1e-12 * scale) ? d[a] * (G[a * 3 + b] - G[b * 3 + a]) / den : 0.0;
        }
    // gW = U Xb V^T
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) T1[a * 3 + b] = U[a * 3 + 0] * Xb[0 * 3 + b] + U[a * 3 + 1] * Xb[1 * 3 + b] + U[a * 3 + 2] * Xb[2 * 3 + b];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) gW[a * 3 + b] = T1[a * 3 + 0] * V[b * 3 + 0] + T1[a * 3 + 1] * V[b * 3 + 1] + T1[a * 3 + 2] * V[b * 3 + 2];
    // W = M/S0 - mut mus^T ; mus = sp/S0 ; mut = sy/S0
    double gS0 = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            gacc[KAB_M + a * 3 + b] = gW[a * 3 + b] / S0;
            gmut[a] -= gW[a * 3 + b] * mus[b];
            gmus[b] -= gW[a * 3 + b] * mut[a];
        }
    // <gW, M>/S0^2 with M/S0 = W + mut mus^T and W = U S V^T (lam_i d_i = s_i)
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double Wab = 0.0;
            for (int k = 0; k < 3; ++k) Wab += U[a * 3 + k] * (lam[k] * d[k]) * V[b * 3 + k];
            gS0 -= gW[a * 3 + b] * (Wab + mut[a] * mus[b]) / S0;
        }
    for (int a = 0; a < 3; ++a) {
        gacc[KAB_SP + a] = gmus[a] / S0;
        gacc[KAB_SY + a] = gmut[a] / S0;
        gS0 -= (gmus[a] * mus[a] + gmut[a] * mut[a]) / S0;
    }
    gacc[KAB_S0] = gS0;
}

}  // namespace dicp
