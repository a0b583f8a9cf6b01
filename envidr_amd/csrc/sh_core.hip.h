// Real spherical harmonics up to degree 8 and the integrated directional encoding (IDE), shared by
// the standalone encoders and the fused render kernel.  Coefficients come from
// sh_ide_tables.inc (generated from the closed forms by tools/gen_tables.py).
#pragma once
#include "common.hip.h"
#include "sh_ide_tables.inc"
#include <utility>

namespace envidr {

// (x + i y)^m for m = 0..M-1 by repeated complex multiplication
template <int M, typename T>
__device__ __forceinline__ void complex_powers(T x, T y, T (&re)[M], T (&im)[M]) {
    re[0] = 1; im[0] = 0;
#pragma unroll
    for (int m = 1; m < M; ++m) {
        re[m] = re[m - 1] * x - im[m - 1] * y;
        im[m] = re[m - 1] * y + im[m - 1] * x;
    }
}

// Y[l*l + l + m] = N Q_l^|m|(z) * (m >= 0 ? Re : Im)((x+iy)^|m|); optional gradient rows dx, dy, dz.
// DEG is the reference's "degree" C (1..8): DEG*DEG outputs.
template <int DEG, bool GRAD>
__device__ __forceinline__ void sh_eval(float x, float y, float z, float* __restrict__ out, float* __restrict__ gx,
                                        float* __restrict__ gy, float* __restrict__ gz) {
    float re[DEG], im[DEG];
    complex_powers<DEG, float>(x, y, re, im);
#pragma unroll
    for (int l = 0; l < DEG; ++l) {
#pragma unroll
        for (int m = 0; m <= l; ++m) {
            // Horner on the (l - m + 1) coefficients; the table is constexpr so they become immediates
            float q = 0, dq = 0;
#pragma unroll
            for (int k = l - m; k >= 0; --k) q = q * z + kShPoly[l][m][k];
            if constexpr (GRAD) {
#pragma unroll
                for (int k = l - m; k >= 1; --k) dq = dq * z + kShPoly[l][m][k] * (float)k;
            }
            const int ip = l * l + l + m, in = l * l + l - m;
            out[ip] = q * re[m];
            if (m) out[in] = q * im[m];
            if constexpr (GRAD) {
                const float mre = m ? (float)m * re[m ? m - 1 : 0] : 0.0f;   // d/dx Re = m Re^(m-1); d/dy Im = m Re^(m-1)
                const float mim = m ? (float)m * im[m ? m - 1 : 0] : 0.0f;   // d/dx Im = m Im^(m-1); d/dy Re = -m Im^(m-1)
                gx[ip] = q * mre;  gy[ip] = q * -mim;  gz[ip] = dq * re[m];
                if (m) { gx[in] = q * mim;  gy[in] = q * mre;  gz[in] = dq * im[m]; }
            }
        }
    }
}

// IDE (ide_encoder/ide_encoder.py:98-130): for i < DEG_VIEW, l = 2^i, m = 0..l:
//   (x+iy)^m * P(z) * exp(-l(l+1)/2 * kappa_inv),   P(z) = sum_k mat[k,(l,m)] z^k
// The reference evaluates P in fp32 where the l = 16 coefficients reach 9e4 and cancel: its own
// result carries up to ~1e-2 of rounding noise near |z| = 1 (DESIGN.md "IDE numerics").  We evaluate
// the SAME fp32-rounded coefficient table in fp64 Horner (in z^2, the polynomials have parity), which
// costs ~120 DFMA per call and is exact to fp32 rounding.
// emit(j, re_part, im_part) receives term j = 0 .. n_terms-1 in the reference's (l, m) order.
// An fp64 constant as two scalar moves issued where it is used.  Left to itself the compiler hoists the 121 coefficient
// pairs out of the kernels' loops, runs out of scalar registers and parks them in VGPR lanes: ~350 v_writelane / v_readlane
// per call, vector-ALU instructions all of them (a third of the call).  `asm volatile` can be neither hoisted nor merged.
template <uint64_t BITS>
__device__ __forceinline__ double sgpr_f64() {
    uint32_t lo, hi;
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"((int)(uint32_t)(BITS & 0xffffffffull)));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"((int)(uint32_t)(BITS >> 32)));
    return __hiloint2double((int)hi, (int)lo);
}
// PIN: the asm form above (the fused kernels, where the call sits inside loops); without it a plain constant -- in the
// standalone operator there is no loop to hoist out of and many waves per SIMD, which the pinned scalar moves only slow down
template <int IDX, bool PIN>
__device__ __forceinline__ double ide_coef() {
    if constexpr (PIN) return sgpr_f64<__builtin_bit_cast(uint64_t, kIdeCoef[IDX])>();
    else return kIdeCoef[IDX];
}
// P(z) = sum_k coef[START + k] z^(D - 2k) as a sum of (scalar coefficient) x (power of z) fused multiply-adds.  The
// coefficient is a MULTIPLICAND on purpose: as the addend of a Horner step a scalar operand has to be copied into the vector
// register the fused multiply-add accumulates in (two v_mov per coefficient).
template <int START, int K, int CNT, int D, int LMAX, bool PIN>
__device__ __forceinline__ double ide_poly(const double p, const double (&zp)[LMAX + 1]) {
    if constexpr (K < CNT) {
        constexpr int n = D - 2 * K;
        if constexpr (n == 0) return ide_poly<START, K + 1, CNT, D, LMAX, PIN>(__builtin_fma(ide_coef<START + K, PIN>(), 1.0, p), zp);
        else return ide_poly<START, K + 1, CNT, D, LMAX, PIN>(__builtin_fma(ide_coef<START + K, PIN>(), zp[n], p), zp);
    } else return p;
}
// (x + i y)^m for m = 0..M-1 in fp64, two multiplies + two fused multiply-adds per power
template <int M>
__device__ __forceinline__ void complex_powers_f64(double x, double y, double (&re)[M], double (&im)[M]) {
    re[0] = 1; im[0] = 0;
    if constexpr (M > 1) { re[1] = x; im[1] = y; }
#pragma unroll
    for (int m = 2; m < M; ++m) {
        re[m] = __builtin_fma(re[m - 1], x, -(im[m - 1] * y));
        im[m] = __builtin_fma(re[m - 1], y, im[m - 1] * x);
    }
}
template <int I, int M, int LMAX, bool PIN, typename Emit>
__device__ __forceinline__ void ide_term(const double (&re)[LMAX + 1], const double (&im)[LMAX + 1], const double (&zp)[LMAX + 1], const float att,
                                         Emit&& emit) {
    constexpr int l = 1 << I, j = l - 1 + I + M, start = kIdeStart[j], cnt = kIdeCount[j], d = l - M;
    static_assert(cnt == d / 2 + 1, "IDE table: a polynomial of degree l - m with parity");
    if constexpr (d == 0) {
        emit(j, (float)(re[M] * ide_coef<start, PIN>()) * att, (float)(im[M] * ide_coef<start, PIN>()) * att);
    } else {
        const double p = ide_poly<start, 1, cnt, d, LMAX, PIN>(ide_coef<start, PIN>() * zp[d], zp);
        emit(j, (float)(re[M] * p) * att, (float)(im[M] * p) * att);
    }
}
template <int I, int LMAX, bool PIN, typename Emit>
__device__ __forceinline__ void ide_level(const double (&re)[LMAX + 1], const double (&im)[LMAX + 1], const double (&zp)[LMAX + 1],
                                          const float kappa_inv, Emit&& emit) {
    constexpr int l = 1 << I;
    const float att = expf(-(0.5f * (float)(l * (l + 1))) * kappa_inv);
    [&]<int... M>(std::integer_sequence<int, M...>) {
        (ide_term<I, M, LMAX, PIN>(re, im, zp, att, emit), ...);
    }(std::make_integer_sequence<int, l + 1>{});
}
template <int DEG_VIEW, bool PIN = false, typename Emit>
__device__ __forceinline__ void ide_eval(float xf, float yf, float zf, float kappa_inv, Emit&& emit) {
    constexpr int LMAX = 1 << (DEG_VIEW - 1);
    double x = xf, y = yf;
    const double z = zf;
    if (xf == 0.0f && yf == 0.0f) y += 1.0;   // reference: y = y + (x == 0 & y == 0)
    double re[LMAX + 1], im[LMAX + 1];
    complex_powers_f64<LMAX + 1>(x, y, re, im);
    double zp[LMAX + 1];                      // z^n
    zp[0] = 1; zp[1] = z;
#pragma unroll
    for (int n = 2; n <= LMAX; ++n) zp[n] = (n & 1) ? zp[n - 1] * z : zp[n / 2] * zp[n / 2];
    [&]<int... I>(std::integer_sequence<int, I...>) {
        (ide_level<I, LMAX, PIN>(re, im, zp, kappa_inv, emit), ...);
    }(std::make_integer_sequence<int, DEG_VIEW>{});
}

__host__ __device__ constexpr int ide_terms(int deg_view) { return (1 << deg_view) - 1 + deg_view; }

// Gradient of the encoding (the reference gets it from torch autograd through ide_encoder.py:98-130, which its training
// branch differentiates: colours -> IDE(reflected direction, roughness) -> normals / roughness head).  Term j is
// v = c^m P(z) att, c = x + i y, P the table polynomial, att = exp(-sigma_l kappa_inv); with upstream gradients (gre, gim)
// of its real and imaginary outputs:
//   d/dx = m P att (gre Re c^(m-1) + gim Im c^(m-1)),   d/dy = m P att (gim Re c^(m-1) - gre Im c^(m-1)),
//   d/dz = P'(z) att (gre Re c^m + gim Im c^m),          d/dkappa_inv = -sigma_l (gre Re v + gim Im v).
// grad(j, gre, gim) supplies the upstream pair of term j.  Same fp64 Horner on the fp32-rounded table as ide_eval.
template <int DEG_VIEW, typename Grad>
__device__ __forceinline__ void ide_grad(float xf, float yf, float zf, float kappa_inv, Grad&& grad, float (&gdir)[3], float& gkappa) {
    constexpr int LMAX = 1 << (DEG_VIEW - 1);
    double x = xf, y = yf;
    const double z = zf;
    if (xf == 0.0f && yf == 0.0f) y += 1.0;
    double re[LMAX + 1], im[LMAX + 1];
    complex_powers<LMAX + 1, double>(x, y, re, im);
    const double z2 = z * z;
    double gx = 0, gy = 0, gz = 0, gk = 0;
    int j = 0;
#pragma unroll
    for (int i = 0; i < DEG_VIEW; ++i) {
        const int l = 1 << i;
        const float sigma = 0.5f * (float)(l * (l + 1));
        const double att = (double)expf(-sigma * kappa_inv);
#pragma unroll
        for (int m = 0; m <= l; ++m, ++j) {
            const int start = kIdeStart[j], cnt = kIdeCount[j];
            double q = kIdeCoef[start], dq = 0;              // q(z^2) and q'(z^2), Horner
#pragma unroll
            for (int k = 1; k < cnt; ++k) { dq = dq * z2 + q; q = q * z2 + kIdeCoef[start + k]; }
            const bool odd = (l - m) & 1;
            const double P = odd ? q * z : q;
            const double dP = odd ? q + 2 * z2 * dq : 2 * z * dq;
            float gre, gim;
            grad(j, gre, gim);
            const double a = gre, b = gim;
            if (m > 0) {
                const double s = (double)m * P * att;
                gx += s * (a * re[m - 1] + b * im[m - 1]);
                gy += s * (b * re[m - 1] - a * im[m - 1]);
            }
            const double w = a * re[m] + b * im[m];
            gz += dP * att * w;
            gk -= (double)sigma * P * att * w;
        }
    }
    gdir[0] = (float)gx; gdir[1] = (float)gy; gdir[2] = (float)gz;
    gkappa = (float)gk;
}

}  // namespace envidr
