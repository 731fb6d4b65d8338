// AmoebaMultipoleForce with PME on the OpenMM "HIP" platform (include/openmm_hip_amoeba.h): permanent multipoles up to quadrupoles,
// induced dipoles (direct polarization), Ewald-split, double precision on the atom side and the platform's float32 grids + FFT.
//
// Oracle: AmoebaReferencePmeMultipoleForce (plugins/amoeba/platforms/reference/src/SimTKReference/AmoebaReferenceMultipoleForce.cpp;
// the functions are cited where they are restated).  The pair arithmetic is NOT the Reference's: that one rotates every pair into a
// quasi-internal frame and works with spherical harmonics (:6335-6751).  Here every pair term is written as
//
//     W(A, B; f) = L_A L_B f(|r|),   r = r_B - r_A,   L_A = q_A - mu_A . grad + Q_A : grad grad,   L_B = q_B + mu_B . grad + Q_B : grad grad
//
// for a radial kernel f that enters only through the chain B_0 = f, B_(n+1) = -(1/r) dB_n/dr:
//     grad^n f  =  sum over pairings of  (+-) B_k  x  (products of r and Kronecker deltas),
// and one device function (mpole_pair) returns W, the force on site A and the torque on A's multipoles for ANY chain.  AMOEBA then
// needs three chains per pair, all consistent derivative chains of their own scalar kernels:
//     m:  erfc(alpha r)/r  -  (1 - m_ij) / r                                  permanent  x  permanent
//     p:  erfc(alpha r)/r  -  (1 - p_ij lambda(r)) / r   (lambda: Thole)      permanent  x  induced dipoles "d"   (p-scaled field)
//     d:  the same with d_ij                                                 permanent  x  induced dipoles "p"
// With direct polarization mu_d = alpha E_d, mu_p = alpha E_p (fields of the permanent multipoles through chains d / p, plus the
// reciprocal field and the self term) the polarization energy -1/2 sum mu_d . E_p equals 1/2 sum_pairs (W1 + W2 + W3 + W4) with
//     W1 = W(mu_d,i / 2, M_j; p)   W2 = W(mu_p,i / 2, M_j; d)   W3 = W(M_i, mu_d,j / 2; p)   W4 = W(M_i, mu_p,j / 2; d)
// (the form the Reference sums, :6456-6751), and its gradient at fixed dipoles is that of W1 + ... + W4 (no factor 1/2):
//     dU = -1/2 sum (mu_p . dE_d + mu_d . dE_p).
//
// Wave64 formulation: per-atom pair lists, rebuilt at every evaluation by a scan that only tests distances (amoeba_pairs.h: the platform's
// slot order, 128-slot tiles with bounding boxes, far tiles skipped), and pair kernels -- fixed field, induced-dipole field, forces -- in
// which MP_SPLIT lanes share an atom and walk its list, so that (nearly) every lane of every iteration holds a pair inside the cutoff.
// Every pair is seen from both of its atoms: a lane group accumulates field / force / torque of its own atom only and the loops hold no
// atomics.  (The first version scanned all atoms from every atom with the pair arithmetic inline: a wavefront then executes that
// arithmetic whenever ANY lane has the candidate inside the cutoff -- 254 ms per step at 36 k atoms against 6.)  Mutual polarization:
// conjugate gradients with device-side step lengths, a per-pair cache of the damped dipole-dipole coefficients for the iterations, induced
// dipoles spread through LDS bricks, the two dipole sets through the FFT side by side on twin grids.
#include "common.h"
#include "../../../include/openmm_hip_amoeba.h"
#include "../../../include/openmm_hip_kernels.h"
#include "amoeba_pairs.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

using namespace omm;

namespace {

#define MP_BLOCK 128
#define MP_CG_BLOCK 256          // the vector kernels of the solver (k_mp_cg)
#define MP_SQRT_PI 1.77245385090551602730

// Small vector algebra, templated on the scalar: double for the atom-side arithmetic, float for the pair arithmetic of the mixed-precision
// kernels (the reference's GPU platforms compute AMOEBA pairs in `real` = float in their mixed mode too and accumulate in fixed point,
// plugins/amoeba/platforms/common/src/kernels/multipoleElectrostatics.cc).
template <class T> struct V3T { T x, y, z; };
typedef V3T<double> V3;
template <class T> __device__ __forceinline__ V3T<T> v3t(T x, T y, T z) { V3T<T> r = {x, y, z}; return r; }
__device__ __forceinline__ V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
template <class T> __device__ __forceinline__ V3T<T> operator+(V3T<T> a, V3T<T> b) { return v3t<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> __device__ __forceinline__ V3T<T> operator-(V3T<T> a, V3T<T> b) { return v3t<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> __device__ __forceinline__ V3T<T> operator*(V3T<T> a, T s) { return v3t<T>(a.x * s, a.y * s, a.z * s); }
template <class T> __device__ __forceinline__ V3T<T> operator*(T s, V3T<T> a) { return v3t<T>(a.x * s, a.y * s, a.z * s); }
template <class T> __device__ __forceinline__ T dot(V3T<T> a, V3T<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> __device__ __forceinline__ V3T<T> cross(V3T<T> a, V3T<T> b) { return v3t<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ double normalize(V3& a) { const double n = sqrt(dot(a, a)); const double inv = n > 0 ? 1.0 / n : 0.0; a = a * inv; return n; }
template <class T> __device__ __forceinline__ V3T<T> convert3(V3 a) { return v3t<T>((T) a.x, (T) a.y, (T) a.z); }
template <class T> __device__ __forceinline__ V3 widen3(V3T<T> a) { return v3((double) a.x, (double) a.y, (double) a.z); }

// symmetric 3x3 as (xx, xy, xz, yy, yz, zz)
template <class T> struct SymT { T xx, xy, xz, yy, yz, zz; };
typedef SymT<double> Sym;
template <class T> __device__ __forceinline__ V3T<T> mul(const SymT<T>& q, V3T<T> v) { return v3t<T>(q.xx * v.x + q.xy * v.y + q.xz * v.z, q.xy * v.x + q.yy * v.y + q.yz * v.z, q.xz * v.x + q.yz * v.y + q.zz * v.z); }
template <class T> __device__ __forceinline__ T ddot(const SymT<T>& a, const SymT<T>& b) { return a.xx * b.xx + a.yy * b.yy + a.zz * b.zz + T(2) * (a.xy * b.xy + a.xz * b.xz + a.yz * b.yz); }
// antisymmetric part of the product A B of two symmetric matrices as a vector: ((AB)_yz - (AB)_zy, (AB)_zx - (AB)_xz, (AB)_xy - (AB)_yx)
template <class T> __device__ __forceinline__ V3T<T> asym_product(const SymT<T>& a, const SymT<T>& b) {
    const T yz = a.xy * b.xz + a.yy * b.yz + a.yz * b.zz, zy = a.xz * b.xy + a.yz * b.yy + a.zz * b.yz;
    const T zx = a.xz * b.xx + a.yz * b.xy + a.zz * b.xz, xz = a.xx * b.xz + a.xy * b.yz + a.xz * b.zz;
    const T xy = a.xx * b.xy + a.xy * b.yy + a.xz * b.yz, yx = a.xy * b.xx + a.yy * b.xy + a.yz * b.xz;
    return v3t<T>(yz - zy, zx - xz, xy - yx);
}
template <class T> __device__ __forceinline__ SymT<T> convert6(const Sym& q) { SymT<T> r = {(T) q.xx, (T) q.xy, (T) q.xz, (T) q.yy, (T) q.yz, (T) q.zz}; return r; }

template <class T> struct SiteT { T q; V3T<T> mu; SymT<T> Q; };
typedef SiteT<double> Site;
template <class T> __device__ __forceinline__ SiteT<T> convert_site(const Site& s) { SiteT<T> r = {(T) s.q, convert3<T>(s.mu), convert6<T>(s.Q)}; return r; }

// a derivative chain with a scale factor s and Thole damping lambda = 1 - oml:  bn - (1 - s lambda) cn  (pair_chains below makes bn, cn, oml)
template <class T> __device__ __forceinline__ T scaled_chain(T bn, T cn, T oml, T s) { return bn - ((T(1) - s) + s * oml) * cn; }

// W = L_A L_B f, the force on site A (= -dW/dr_A) and the torque on A's multipoles, for the kernel given by its chain B[0..5].
// Quadrupoles are traceless (AMOEBA's are: rotations of a traceless local-frame tensor).
// AQ / BQ = false: the site is a bare dipole (an induced one: q = 0, Q = 0; no torque is returned for such an A) -- the terms that
// vanish are not compiled; four of the five to seven calls per pair of the force kernel have a bare dipole on one side.
template <class T, bool AQ, bool BQ>
__device__ __forceinline__ void mpole_pair(const SiteT<T>& A, const SiteT<T>& Bs, const V3T<T> r, const T* B, T& W, V3T<T>& force, V3T<T>& torque) {
    const T two = T(2), four = T(4);
    const T muAr = dot(A.mu, r), muBr = dot(Bs.mu, r), muAmuB = dot(A.mu, Bs.mu);
    V3T<T> QAr = v3t<T>(0, 0, 0), QBr = v3t<T>(0, 0, 0);
    T rQAr = T(0), rQBr = T(0);
    if constexpr (AQ) { QAr = mul(A.Q, r); rQAr = dot(r, QAr); }
    if constexpr (BQ) { QBr = mul(Bs.Q, r); rQBr = dot(r, QBr); }
    T S1 = B[2] * muBr, S2 = -B[3] * muBr;
    if constexpr (BQ) { S1 += -Bs.q * B[1] - B[3] * rQBr; S2 += Bs.q * B[2] + B[4] * rQBr; }
    V3T<T> gradPhi = S1 * r - B[1] * Bs.mu;                                  // the field of B at A (r-derivative of its potential)
    if constexpr (BQ) gradPhi = gradPhi + (two * B[2]) * QBr;
    W = -dot(A.mu, gradPhi);
    // mu_A . Hessian(phi)
    V3T<T> muAH = (S2 * muAr + B[2] * muAmuB) * r + (B[2] * muAr) * Bs.mu + S1 * A.mu;
    if constexpr (BQ) {
        const T muAQBr = dot(A.mu, QBr);
        muAH = muAH - (two * B[3] * muAQBr) * r - (two * B[3] * muAr) * QBr + (two * B[2]) * mul(Bs.Q, A.mu);
    }
    force = v3t<T>(0, 0, 0) - muAH;
    torque = v3t<T>(0, 0, 0);
    if constexpr (AQ) {
        T S3 = B[4] * muBr, phi = -B[1] * muBr;
        if constexpr (BQ) { S3 += -Bs.q * B[3] - B[5] * rQBr; phi += Bs.q * B[0] + B[2] * rQBr; }
        const T muBQAr = dot(Bs.mu, QAr);
        const V3T<T> QAmuB = mul(A.Q, Bs.mu);
        W += A.q * phi + S2 * rQAr + two * B[2] * muBQAr;
        // Q_A : third derivatives of phi
        V3T<T> QAD = (S3 * rQAr - two * B[3] * muBQAr) * r - (B[3] * rQAr) * Bs.mu + (two * S2) * QAr + (two * B[2]) * QAmuB;
        V3T<T> tq = S2 * cross(QAr, r) + B[2] * (cross(QAmuB, r) + cross(QAr, Bs.mu));
        if constexpr (BQ) {
            const T QArQBr = dot(QAr, QBr), QAQB = ddot(A.Q, Bs.Q);
            const V3T<T> QAQBr = mul(A.Q, QBr), QBQAr = mul(Bs.Q, QAr);
            W += -four * B[3] * QArQBr + two * B[2] * QAQB;
            QAD = QAD + (four * B[4] * QArQBr - two * B[3] * QAQB) * r + (two * B[4] * rQAr) * QBr - (four * B[3]) * (QBQAr + QAQBr);
            tq = tq - (two * B[3]) * (cross(QAQBr, r) + cross(QAr, QBr)) + (two * B[2]) * asym_product(A.Q, Bs.Q);
        }
        force = force + A.q * gradPhi + QAD;
        torque = cross(A.mu, gradPhi) - two * tq;
    }
}

// One pair of the force kernel, seen from atom i: permanent x permanent through chain m, permanent x induced through chains p and d, and
// (mutual polarization) induced x induced.  Returns this pair's energy, force on i and torque on i's multipoles WITHOUT the Coulomb
// constant.  udI ... upJ are the induced dipoles themselves (the halves of the energy expression are applied here).
template <class T>
__device__ __forceinline__ void forces_pair(const SiteT<T>& Mi, const SiteT<T>& Mj, V3T<T> udI, V3T<T> upI, V3T<T> udJ, V3T<T> upJ, V3T<T> r,
                                            const T (&bn)[6], const T (&cn)[6], const T (&oml)[5], T scM, T scP, T scD, bool mutual,
                                            T& energy, V3T<T>& force, V3T<T>& torque) {
    const T half = T(0.5), quarter = T(0.25);
    const SymT<T> zeroQ = {T(0), T(0), T(0), T(0), T(0), T(0)};
    const SiteT<T> halfUdI = {T(0), half * udI, zeroQ}, halfUpI = {T(0), half * upI, zeroQ}, halfUdJ = {T(0), half * udJ, zeroQ}, halfUpJ = {T(0), half * upJ, zeroQ};
    T chain[6], W; V3T<T> f, tq;
    // permanent x permanent
    for (int n = 0; n < 6; n++) chain[n] = bn[n] - (T(1) - scM) * cn[n];
    mpole_pair<T, true, true>(Mi, Mj, r, chain, W, f, tq);
    energy = half * W; force = f; torque = tq;
    // permanent x induced: chain p with the "d" dipoles, chain d with the "p" dipoles
    chain[0] = T(0); chain[5] = T(0);
    for (int n = 1; n < 5; n++) chain[n] = scaled_chain(bn[n], cn[n], oml[n], scP);
    mpole_pair<T, false, true>(halfUdI, Mj, r, chain, W, f, tq);          // W1: my induced dipole in j's permanent field (no torque: induced dipoles have no frame)
    energy += quarter * W; force = force + f;
    mpole_pair<T, true, false>(Mi, halfUdJ, r, chain, W, f, tq);          // W3: my permanent multipoles in the field of j's induced dipole
    energy += quarter * W; force = force + f; torque = torque + tq;
    for (int n = 1; n < 5; n++) chain[n] = scaled_chain(bn[n], cn[n], oml[n], scD);
    mpole_pair<T, false, true>(halfUpI, Mj, r, chain, W, f, tq);          // W2
    energy += quarter * W; force = force + f;
    mpole_pair<T, true, false>(Mi, halfUpJ, r, chain, W, f, tq);          // W4
    energy += quarter * W; force = force + f; torque = torque + tq;
    if (mutual) {
        // the induced dipoles polarize each other: -1/2 mu_d (dT/dx) mu_p, the gradient of W(mu_d,i, mu_p,j) / 2 + W(mu_p,i, mu_d,j) / 2
        // at fixed dipoles through the Thole-damped chain (u scale = 1, as in calculateDirectInducedDipolePairIxns :6172-6230)
        for (int n = 1; n < 5; n++) chain[n] = bn[n] - oml[n] * cn[n];
        mpole_pair<T, false, false>(halfUdI, halfUpJ, r, chain, W, f, tq);
        force = force + T(2) * f;
        mpole_pair<T, false, false>(halfUpI, halfUdJ, r, chain, W, f, tq);
        force = force + T(2) * f;
    }
}

// The field of partner j's permanent multipoles at atom i through the chains d and p (calculateFixedMultipoleFieldPairIxn :5079-5167):
// field = S1 r - B1 mu + 2 B2 Q r  with  S1 = -q B1 + B2 (mu.r) - B3 (r.Q.r)
template <class T>
__device__ __forceinline__ void field_pair(const SiteT<T>& Mj, V3T<T> r, const T (&bn)[6], const T (&cn)[6], const T (&oml)[5], T scD, T scP, V3T<T>& ed, V3T<T>& ep) {
    const V3T<T> Qr = mul(Mj.Q, r);
    const T mur = dot(Mj.mu, r), rQr = dot(r, Qr);
    T b1 = scaled_chain(bn[1], cn[1], oml[1], scD), b2 = scaled_chain(bn[2], cn[2], oml[2], scD), b3 = scaled_chain(bn[3], cn[3], oml[3], scD);
    ed = (-Mj.q * b1 + b2 * mur - b3 * rQr) * r - b1 * Mj.mu + (T(2) * b2) * Qr;
    b1 = scaled_chain(bn[1], cn[1], oml[1], scP); b2 = scaled_chain(bn[2], cn[2], oml[2], scP); b3 = scaled_chain(bn[3], cn[3], oml[3], scP);
    ep = (-Mj.q * b1 + b2 * mur - b3 * rQr) * r - b1 * Mj.mu + (T(2) * b2) * Qr;
}

struct MpArgs {
    int n, paddedAtoms, includeEnergy, energySlots, nx, ny, nz;
    const double4* pos;
    const double* charge; const double* molDipole; const double* molQuad; const int4* axis;
    const double* thole; const double* damping; const double* polarity;
    const int* specStart; const int* specAtom; const double4* specScale;
    double cutoff2, alpha;
    double* labDipole; double* labQuad; double* fieldD; double* fieldP; double* indD; double* indP; double* phi; double* phiInd; double* torque;
    int mutual;
    double* phiIndP;           // mutual: potential of mu_p (phiInd then holds that of mu_d)
    BoxD box;
    double a[3][3];            // a[k][c] = d(grid coordinate k) / d(Cartesian c) = n_k * recip[c][k]
    float* grid;
    float* grid2;                                  // second grid of a two-grid launch (blockIdx.y == 1), else unused
    int clearGrids;                                // k_mp_cg stage 3 also zeroes grid and grid2 (for the spreading of the next iteration)
    const int* slotOfAtom;
    omm_fixed* force;
    double* energyBuffer;
    // pair lists (amoeba_pairs.h), built once per evaluation: thread g of the pair kernels owns the atom at scan position g -- order[g], the
    // platform's slot order (-1: padding), or g itself without an order -- and walks pairList[k * listStride + g], k < pairCount[g]
    const int* order; int numScan;
    const int* pairList; const int* pairCount; int listStride, listSubcap;
    int precond; double precondCut2;               // neighbour-pair preconditioner of the solver (needs the pair cache)
    const double* needDone;                        // kernels of a tail enqueued before the host has seen the convergence word run only if it is set (solve_mutual)
    const int* listOverflow; const int* listBuilds; // the list builder's overflow word and build counter (device): k_mp_cg stage 5 copies them into sums[13], sums[15]
    double* torque2;                               // k_mp_special<true> leaves its torques HERE (it runs beside k_mp_forces<true>, which stores into torque); k_mp_torque_to_force adds the two
    int pairsOnly;                                 // k_mp_field stops after the pair sums (k_mp_field_finish follows behind the wait for the reciprocal potential)
    int specialAdds;                               // mixed precision: k_mp_forces<true> ran first and STORED its torques, k_mp_special<true> adds to them (0: the other way round)
    float* pairCache; int pairCap;                 // mutual polarization: per list entry (dx, dy, dz, b1, b2) of the Thole-damped dipole-dipole chain, float planes of pairCap * listStride
    float* gather;                                 // mutual polarization: the vectors the induced-dipole field is taken of, (vD, vP) as six floats per SCAN POSITION (k_mp_dipole_field gathers them)
    const double* doneFlag;                        // mutual polarization: sums[10] of the solver -- non-zero once the dipoles have converged: kernels of iterations enqueued ahead return at once
                                                   // (also when doneFlag[3] = sums[13], the list builder's overflow word, is set: that solve is thrown away by the host at its first wait)
    const double4* specScaleSorted;                // scale factors of the special pairs, rows in the order the list entries index them
};

// atom at scan position g (-1: none)
__device__ __forceinline__ int scan_atom(const MpArgs& a, int g) { return g < a.numScan ? (a.order != nullptr ? a.order[g] : g) : -1; }

__device__ __forceinline__ V3 load3(const double* p, int i) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ void store3(double* p, int i, V3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
__device__ __forceinline__ Sym load6(const double* p, int i) { Sym q = {p[6 * i], p[6 * i + 1], p[6 * i + 2], p[6 * i + 3], p[6 * i + 4], p[6 * i + 5]}; return q; }
__device__ __forceinline__ V3 position(const MpArgs& a, int i) { const double4 p = a.pos[i]; return v3(p.x, p.y, p.z); }

// ------------------------------------------------------------------------------------------------
// Local frames -> lab-frame multipoles.  AmoebaReferenceMultipoleForce::checkChiralCenterAtParticle (:354-378) and
// applyRotationMatrixToParticle (:396-503); positions are used as they are (molecules are whole), as there.
// axis types (AmoebaMultipoleForce::MultipoleAxisTypes): 0 ZThenX, 1 Bisector, 2 ZBisect, 3 ThreeFold, 4 ZOnly, 5 NoAxisType
// ------------------------------------------------------------------------------------------------
__global__ void k_mp_frames(MpArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int4 ax = a.axis[i];
    V3 d = load3(a.molDipole, i);
    Sym q = load6(a.molQuad, i);
    store3(a.torque, i, v3(0, 0, 0));
    if (ax.x == 5 || ax.y < 0) { store3(a.labDipole, i, d); for (int k = 0; k < 6; k++) a.labQuad[6 * i + k] = a.molQuad[6 * i + k]; return; }
    const V3 xi = position(a, i);
    if (ax.x == 0 && ax.w >= 0 && ax.z >= 0) {
        // an inverted chiral centre flips the y components
        const V3 py = position(a, ax.w);
        const V3 ad = xi - py, bd = position(a, ax.y) - py, cd = position(a, ax.z) - py;
        if (dot(cross(bd, cd), ad) < 0.0) { d.y = -d.y; q.xy = -q.xy; q.yz = -q.yz; }
    }
    V3 vz = position(a, ax.y) - xi, vx, vy;
    normalize(vz);
    if (ax.x == 4) vx = fabs(vz.x) < 0.866 ? v3(1, 0, 0) : v3(0, 1, 0);
    else {
        vx = position(a, ax.z) - xi;
        if (ax.x == 1) { normalize(vx); vz = vz + vx; normalize(vz); }
        else if (ax.x == 2) { normalize(vx); vy = position(a, ax.w) - xi; normalize(vy); vx = vx + vy; normalize(vx); }
        else if (ax.x == 3) { normalize(vx); vy = position(a, ax.w) - xi; normalize(vy); vz = vz + vx + vy; normalize(vz); }
    }
    vx = vx - vz * dot(vz, vx);
    normalize(vx);
    vy = cross(vz, vx);
    // lab = R^T local with rows of R = (vx, vy, vz)
    store3(a.labDipole, i, d.x * vx + d.y * vy + d.z * vz);
    const double R[3][3] = {{vx.x, vx.y, vx.z}, {vy.x, vy.y, vy.z}, {vz.x, vz.y, vz.z}};
    const double m[3][3] = {{q.xx, q.xy, q.xz}, {q.xy, q.yy, q.yz}, {q.xz, q.yz, q.zz}};
    double lab[3][3];
    for (int c = 0; c < 3; c++)
        for (int e = c; e < 3; e++) {
            double s = 0;
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) s += R[k][c] * R[l][e] * m[k][l];
            lab[c][e] = s;
        }
    a.labQuad[6 * i] = lab[0][0]; a.labQuad[6 * i + 1] = lab[0][1]; a.labQuad[6 * i + 2] = lab[0][2];
    a.labQuad[6 * i + 3] = lab[1][1]; a.labQuad[6 * i + 4] = lab[1][2]; a.labQuad[6 * i + 5] = lab[2][2];
}

// ------------------------------------------------------------------------------------------------
// Order-5 B-spline weights of one coordinate and their first three derivatives with respect to the grid coordinate.
// Same index convention as pme.hip (bspline): grid points index + 0..4 (mod n).  The k-th derivative of the order-5 spline is the
// k-th backward difference of the order-(5 - k) spline.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void bspline4(double u, int n, int& index, double (&th)[4][5]) {
    const int ti = (int) floor(u);
    const double w = u - ti;
    index = ((ti % n) + n) % n;
    double o2[2] = {1.0 - w, w};
    double o3[3], o4[4], o5[5];
    o3[0] = 0.5 * (1.0 - w) * o2[0]; o3[1] = 0.5 * ((w + 1.0) * o2[0] + (2.0 - w) * o2[1]); o3[2] = 0.5 * w * o2[1];
    o4[0] = (1.0 / 3.0) * (1.0 - w) * o3[0];
    o4[1] = (1.0 / 3.0) * ((w + 2.0) * o3[0] + (2.0 - w) * o3[1]);
    o4[2] = (1.0 / 3.0) * ((w + 1.0) * o3[1] + (3.0 - w) * o3[2]);
    o4[3] = (1.0 / 3.0) * w * o3[2];
    o5[0] = 0.25 * (1.0 - w) * o4[0];
    o5[1] = 0.25 * ((w + 3.0) * o4[0] + (2.0 - w) * o4[1]);
    o5[2] = 0.25 * ((w + 2.0) * o4[1] + (3.0 - w) * o4[2]);
    o5[3] = 0.25 * ((w + 1.0) * o4[2] + (4.0 - w) * o4[3]);
    o5[4] = 0.25 * w * o4[3];
    for (int k = 0; k < 5; k++) {
        th[0][k] = o5[k];
        const double a1 = k >= 1 ? o4[k - 1] : 0.0, a0 = k <= 3 ? o4[k] : 0.0;
        th[1][k] = a1 - a0;
        const double b2 = k >= 2 ? o3[k - 2] : 0.0, b1 = (k >= 1 && k <= 3) ? o3[k - 1] : 0.0, b0 = k <= 2 ? o3[k] : 0.0;
        th[2][k] = b2 - 2.0 * b1 + b0;
        const double c3 = k >= 3 ? o2[k - 3] : 0.0, c2 = (k >= 2 && k <= 3) ? o2[k - 2] : 0.0, c1 = (k >= 1 && k <= 2) ? o2[k - 1] : 0.0, c0 = k <= 1 ? o2[k] : 0.0;
        th[3][k] = c3 - 3.0 * c2 + 3.0 * c1 - c0;
    }
}

__device__ __forceinline__ void atom_splines(const MpArgs& a, V3 x, int (&idx)[3], double (&th)[3][4][5]) {
    // grid coordinates u_k = sum_c a[k][c] x_c, wrapped into [0, n_k)
    const int n[3] = {a.nx, a.ny, a.nz};
    for (int k = 0; k < 3; k++) {
        double u = a.a[k][0] * x.x + a.a[k][1] * x.y + a.a[k][2] * x.z;
        u -= floor(u / n[k]) * n[k];
        bspline4(u, n[k], idx[k], th[k]);
    }
}

// Spreads L_i W(g; r_i) = [q + mu . grad_i + Q : grad_i grad_i] W for the permanent multipoles (INDUCED = false) or the dipoles
// (mu_d + mu_p) / 2 (INDUCED = true) onto the float grid.  AmoebaReferencePmeMultipoleForce::spreadFixedMultipolesOntoGrid (:5380-5423),
// spreadInducedDipolesOnGrid (:5572-5616); derivatives with respect to the atom position through the chain rule a[k][c].
// INDUCED: the dipoles sA * A + sB * B (B may be null)
// Eight lanes per atom: lane l < 5 owns the stencil points with z offset l and walks through the 25 (x, y) offsets, so one atomic
// instruction of a wave adds to 8 runs of 5 consecutive grid cells -- about 12 memory-side transactions (64-byte lines) instead of
// the 64 that one-atom-per-lane costs; the float atomics, not the arithmetic, are what this kernel's time consists of.
#define MP_SPREAD_LANES 8
template <bool INDUCED>
__global__ void k_mp_spread(MpArgs a, const double* __restrict__ A, double sA, const double* __restrict__ B, double sB) {
    if (a.doneFlag != nullptr && (a.doneFlag[0] != 0.0 || a.doneFlag[3] != 0.0)) return;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = tid / MP_SPREAD_LANES, iz = tid % MP_SPREAD_LANES;
    if (i >= a.n || iz >= 5) return;
    int idx[3];
    double th[3][4][5];
    atom_splines(a, position(a, i), idx, th);
    const double q = INDUCED ? 0.0 : a.charge[i];
    const V3 mu = INDUCED ? sA * load3(A, i) + (B != nullptr ? sB * load3(B, i) : v3(0, 0, 0)) : load3(a.labDipole, i);
    // fractional moments: d_k = sum_c a[k][c] mu_c,  Q_kl = sum_cd a[k][c] a[l][d] Q_cd
    double fd[3], fq[3][3];
    const double muc[3] = {mu.x, mu.y, mu.z};
    for (int k = 0; k < 3; k++) fd[k] = a.a[k][0] * muc[0] + a.a[k][1] * muc[1] + a.a[k][2] * muc[2];
    if (!INDUCED) {
        const Sym Q = load6(a.labQuad, i);
        const double m[3][3] = {{Q.xx, Q.xy, Q.xz}, {Q.xy, Q.yy, Q.yz}, {Q.xz, Q.yz, Q.zz}};
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 3; l++) {
                double s = 0;
                for (int c = 0; c < 3; c++)
                    for (int d = 0; d < 3; d++) s += a.a[k][c] * a.a[l][d] * m[c][d];
                fq[k][l] = s;
            }
    }
    // this lane's z weights (value, first and second derivative), picked with selects: a dynamic index would put the table into scratch
    double w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
    for (int z = 0; z < 5; z++) if (z == iz) { w0 = th[2][0][z]; w1 = th[2][1][z]; w2 = th[2][2][z]; }
    const int gz = (idx[2] + iz) % a.nz;
    for (int ix = 0; ix < 5; ix++) {
        const int gx = (idx[0] + ix) % a.nx;
        for (int iy = 0; iy < 5; iy++) {
            const int gy = (idx[1] + iy) % a.ny;
            const double t0 = th[0][0][ix], t1 = th[0][1][ix], t2 = th[0][2][ix], u0 = th[1][0][iy], u1 = th[1][1][iy], u2 = th[1][2][iy];
            double term0, term1, term2 = 0.0;
            if (INDUCED) { term0 = fd[0] * t1 * u0 + fd[1] * t0 * u1; term1 = fd[2] * t0 * u0; }
            else {
                term0 = q * t0 * u0 + fd[0] * t1 * u0 + fd[1] * t0 * u1 + fq[0][0] * t2 * u0 + fq[1][1] * t0 * u2 + 2.0 * fq[0][1] * t1 * u1;
                term1 = fd[2] * t0 * u0 + 2.0 * fq[0][2] * t1 * u0 + 2.0 * fq[1][2] * t0 * u1;
                term2 = fq[2][2] * t0 * u0;
            }
            const double v = term0 * w0 + term1 * w1 + term2 * w2;
            atomicAdd(&a.grid[((size_t) gx * a.ny + gy) * a.nz + gz], (float) v);
        }
    }
}

// Induced dipoles through LDS bricks (the scheme of the platform's charge spreading, pme.hip): one workgroup takes the 32 atoms of a block
// of the slot order -- spatial neighbours, ~7 grid cells across --, accumulates their 125-point stencils in a 16^3 brick held in LDS (32-bit
// fixed point, integer LDS atomics) and flushes the touched cells with z-contiguous float atomics: a few hundred memory-side transactions
// per 32 atoms instead of ~37 per atom.  An atom whose stencil does not fit the brick adds to the grid directly.  Needs the slot order
// (a.order); the dipoles are sA * A + sB * B as in k_mp_spread<true>.
#define MPB_ATOMS 32
#define MPB_BRICK 16
#define MPB_ZS (MPB_BRICK + 1)
__device__ __forceinline__ int mpb_wrap_rel(int d, int n) { if (d >= (n + 1) / 2) d -= n; if (d < -(n / 2)) d += n; return d; }

// cgw (optional, two-grid launches of the solver): the solver's work vectors -- the dipoles to spread are the NEW search direction p = z + b p,
// formed here per atom, stored, and packed for k_mp_dipole_field's gather (k_mp_cg stage 3 as a launch of its own otherwise); b = sums[8 + set],
// left by the last block of stage 7.
__global__ __launch_bounds__(256) void k_mp_spread_bricks(MpArgs a, const double* __restrict__ A, double sA, const double* __restrict__ B, double sB, const double* __restrict__ A2, double* cgw) {
    if (a.doneFlag != nullptr && (a.doneFlag[0] != 0.0 || a.doneFlag[3] != 0.0)) return;           // an iteration enqueued ahead of the convergence check (solve_mutual)
    if (a.needDone != nullptr && *a.needDone == 0.0) return;           // the tail enqueued ahead of it: only once converged
    // two-grid launch: the second set of dipoles onto the second grid (the argument struct itself is left alone: a modified copy would
    // move all of it from scalar kernel-argument loads to private memory)
    float* const grid = blockIdx.y == 1 ? a.grid2 : a.grid;
    if (blockIdx.y == 1) A = A2;
    __shared__ int brick[MPB_BRICK * MPB_BRICK * MPB_ZS];
    __shared__ float th[MPB_ATOMS][3][5], dth[MPB_ATOMS][3][5], fd[MPB_ATOMS][3];
    __shared__ int base[MPB_ATOMS][3], ok[MPB_ATOMS], ref[3], minRel[3];
    __shared__ float sMax;
    const int t = threadIdx.x, g0 = blockIdx.x * MPB_ATOMS;
    const int n[3] = {a.nx, a.ny, a.nz};
    if (t < 3) { minRel[t] = 1 << 30; ref[t] = -1; }
    if (t == 0) sMax = 0.f;
    for (int w = t; w < MPB_BRICK * MPB_BRICK * MPB_ZS; w += 256) brick[w] = 0;
    __syncthreads();
    // splines and fractional dipoles: thread (atom, k), k < 3: dimension k; k == 3: the dipole in grid coordinates
    if (t < 4 * MPB_ATOMS) {
        const int atom = t >> 2, k = t & 3, i = scan_atom(a, g0 + atom);
        if (k == 3) ok[atom] = i >= 0;
        if (i >= 0) {
            if (k < 3) {
                const V3 x = position(a, i);
                double u = a.a[k][0] * x.x + a.a[k][1] * x.y + a.a[k][2] * x.z;
                u -= floor(u / n[k]) * n[k];
                int index; double w4[4][5];
                bspline4(u, n[k], index, w4);
                base[atom][k] = index;
                for (int m = 0; m < 5; m++) { th[atom][k][m] = (float) w4[0][m]; dth[atom][k][m] = (float) w4[1][m]; }
            }
            else {
                V3 mu;
                if (cgw != nullptr) {
                    const size_t n3 = 3 * (size_t) a.n;
                    const int set = blockIdx.y;
                    double* const pv = cgw + (4 + set) * n3;
                    mu = load3(cgw + (2 + set) * n3, i) + cgw[8 * n3 + 8 + set] * load3(pv, i);
                    store3(pv, i, mu);
                    if (a.gather != nullptr) { float* v = a.gather + 6 * (size_t) (g0 + atom) + 3 * set; v[0] = (float) mu.x; v[1] = (float) mu.y; v[2] = (float) mu.z; }
                }
                else mu = sA * load3(A, i) + (B != nullptr ? sB * load3(B, i) : v3(0, 0, 0));
                float sum = 0.f;
                for (int c = 0; c < 3; c++) { const float f = (float) (a.a[c][0] * mu.x + a.a[c][1] * mu.y + a.a[c][2] * mu.z); fd[atom][c] = f; sum += fabsf(f); }
                atomicMax((int*) &sMax, __float_as_int(sum));          // non-negative floats order like their bit patterns
            }
        }
    }
    __syncthreads();
    if (t < 64) {
        const unsigned long long valid = __ballot(t < MPB_ATOMS && ok[t < MPB_ATOMS ? t : 0] != 0);
        if (valid != 0 && t == __ffsll((long long) valid) - 1) { ref[0] = base[t][0]; ref[1] = base[t][1]; ref[2] = base[t][2]; }
    }
    __syncthreads();
    if (ref[0] < 0) return;                                    // no atom in this block (padding)
    if (t < MPB_ATOMS && ok[t])
        for (int d = 0; d < 3; d++) atomicMin(&minRel[d], mpb_wrap_rel(base[t][d] - ref[d], n[d]));
    __syncthreads();
    // fixed-point scale: 32 atoms of the largest |fd| sum stacked on one cell (weights: |theta'| <= 0.7, theta <= 0.6) stay below 2^30
    const float scale = exp2f(floorf(30.f - log2f(fmaxf(MPB_ATOMS * 0.3f * sMax, 1e-30f))));
    const int lane = t & 63, wave = t >> 6;
    for (int kk = 0; kk < MPB_ATOMS / 4; kk++) {
        const int atom = wave * (MPB_ATOMS / 4) + kk;
        if (!ok[atom]) continue;                               // wave-uniform
        int off[3];
        bool fits = true;
        for (int d = 0; d < 3; d++) { off[d] = mpb_wrap_rel(base[atom][d] - ref[d], n[d]) - minRel[d]; fits = fits && off[d] + 5 <= MPB_BRICK && MPB_BRICK <= n[d]; }
        for (int pt = lane; pt < 125; pt += 64) {
            const int ix = pt / 25, iy = (pt / 5) % 5, iz = pt % 5;
            const float v = fd[atom][0] * dth[atom][0][ix] * th[atom][1][iy] * th[atom][2][iz] + fd[atom][1] * th[atom][0][ix] * dth[atom][1][iy] * th[atom][2][iz]
                          + fd[atom][2] * th[atom][0][ix] * th[atom][1][iy] * dth[atom][2][iz];
            if (fits) atomicAdd(&brick[((off[0] + ix) * MPB_BRICK + off[1] + iy) * MPB_ZS + off[2] + iz], __float2int_rn(v * scale));
            else {
                const int gx = (base[atom][0] + ix) % a.nx, gy = (base[atom][1] + iy) % a.ny, gz = (base[atom][2] + iz) % a.nz;
                atomicAdd(&grid[((size_t) gx * a.ny + gy) * a.nz + gz], v);
            }
        }
    }
    __syncthreads();
    int org[3];
    for (int d = 0; d < 3; d++) { org[d] = (ref[d] + minRel[d]) % n[d]; if (org[d] < 0) org[d] += n[d]; }
    const float invScale = 1.f / scale;
    for (int w = t; w < MPB_BRICK * MPB_BRICK * MPB_ZS; w += 256) {
        const int fixed = brick[w];
        if (fixed != 0) {
            int gx = org[0] + w / (MPB_BRICK * MPB_ZS); gx -= gx >= a.nx ? a.nx : 0;
            int gy = org[1] + (w / MPB_ZS) % MPB_BRICK; gy -= gy >= a.ny ? a.ny : 0;
            int gz = org[2] + w % MPB_ZS; gz -= gz >= a.nz ? a.nz : 0;
            atomicAdd(&grid[((size_t) gx * a.ny + gy) * a.nz + gz], (float) fixed * invScale);
        }
    }
}

// The convolved grid back at the atoms: potential and its derivatives up to third order, Cartesian, with respect to the atom position:
//   out[0] phi | [1..3] x y z | [4..9] xx xy xz yy yz zz | [10..19] xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz
// AmoebaReferencePmeMultipoleForce::computeFixedPotentialFromGrid (:5464-5570) / computeInducedPotentialFromGrid (:5618-5818) and
// transformPotentialToCartesianCoordinates (:5340-5378), here with the general chain rule for all three orders.
// Eight lanes per atom, as in the spreading kernel: lane l < 5 reads the stencil points with z offset l (the five lanes of an atom read
// five consecutive grid cells) and sums them over the 25 (x, y) offsets into the ten (p, q) combinations; the z weights of the lane's
// own offset turn those into its share of the 20 derivatives, and the shares meet through shuffles.  (One thread per atom held the
// whole 3 x 4 x 5 weight table and 20 sums in registers and spilled.)
// MAXORD = 3: all 20 values; MAXORD = 1: the potential and its gradient only (out[0..3]) -- all that the induced-dipole field of a solver
// iteration reads, at a third of the arithmetic; MAXORD = 2: up to the second derivatives (out[0..9]), the field gradient the
// extrapolated-polarization scheme keeps of every order.
template <int MAXORD>
__global__ void k_mp_potential(MpArgs a, double* __restrict__ out, double* __restrict__ out2) {
    if (a.doneFlag != nullptr && (a.doneFlag[0] != 0.0 || a.doneFlag[3] != 0.0)) return;
    if (a.needDone != nullptr && *a.needDone == 0.0) return;
    const float* const grid = blockIdx.y == 1 ? a.grid2 : a.grid;          // (see k_mp_spread_bricks)
    if (blockIdx.y == 1) out = out2;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = tid / MP_SPREAD_LANES, iz = tid % MP_SPREAD_LANES;
    const bool atom = i < a.n, mine = atom && iz < 5;
    int idx[3];
    double th[3][4][5];
    atom_splines(a, position(a, atom ? i : 0), idx, th);
    constexpr int NO = MAXORD + 1;
    double G[NO][NO];                  // G[p][q] = sum over (ix, iy) of grid(ix, iy, my z) thx^(p)[ix] thy^(q)[iy],  p + q <= MAXORD
    for (int p = 0; p < NO; p++) for (int q = 0; q < NO; q++) G[p][q] = 0.0;
    if (mine) {
        const int gz = (idx[2] + iz) % a.nz;
        for (int ix = 0; ix < 5; ix++) {
            const int gx = (idx[0] + ix) % a.nx;
            for (int iy = 0; iy < 5; iy++) {
                const int gy = (idx[1] + iy) % a.ny;
                const double g = (double) grid[((size_t) gx * a.ny + gy) * a.nz + gz];
                for (int p = 0; p < NO; p++) {
                    const double gp = g * th[0][p][ix];
                    for (int q = 0; p + q < NO; q++) G[p][q] += gp * th[1][q][iy];
                }
            }
        }
    }
    // this lane's z weights by derivative order (selects, not a dynamic index into the table)
    double wz[4] = {0, 0, 0, 0};
#pragma unroll
    for (int z = 0; z < 5; z++) if (z == iz) { wz[0] = th[2][0][z]; wz[1] = th[2][1][z]; wz[2] = th[2][2][z]; wz[3] = th[2][3][z]; }
    // fractional derivatives F[p][q][r] = sum_g grid(g) thx^(p) thy^(q) thz^(r),  p + q + r <= MAXORD: sum of the lanes' shares
    double F[NO][NO][NO];
    for (int p = 0; p < NO; p++)
        for (int q = 0; p + q < NO; q++)
            for (int r = 0; p + q + r < NO; r++) {
                double v = G[p][q] * wz[r];
                v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
                F[p][q][r] = v;
            }
    if (!atom || iz != 0) return;
    double* o = out + 20 * (size_t) i;
    const double f1[3] = {F[1][0][0], F[0][1][0], F[0][0][1]};
    o[0] = F[0][0][0];
    for (int c = 0; c < 3; c++) o[1 + c] = a.a[0][c] * f1[0] + a.a[1][c] * f1[1] + a.a[2][c] * f1[2];
    if (MAXORD >= 2) {
        // second and third fractional derivative tensors by index
        double f2[3][3], f3[3][3][3];
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 3; l++) {
                int e[3] = {0, 0, 0}; e[k]++; e[l]++;
                f2[k][l] = F[e[0] % NO][e[1] % NO][e[2] % NO];
                if (MAXORD >= 3)
                    for (int m = 0; m < 3; m++) { int e3[3] = {e[0], e[1], e[2]}; e3[m]++; f3[k][l][m] = F[e3[0] % NO][e3[1] % NO][e3[2] % NO]; }
            }
        int n2 = 4, n3 = 10;
        for (int c = 0; c < 3; c++)
            for (int d = c; d < 3; d++) {
                double s2 = 0;
                for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) s2 += a.a[k][c] * a.a[l][d] * f2[k][l];
                o[n2++] = s2;
                if (MAXORD >= 3)
                    for (int e = d; e < 3; e++) {
                        double t3 = 0;
                        for (int k = 0; k < 3; k++) for (int l = 0; l < 3; l++) for (int m = 0; m < 3; m++) t3 += a.a[k][c] * a.a[l][d] * a.a[m][e] * f3[k][l][m];
                        o[n3++] = t3;
                    }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// Pair chains.  bn[0..5]: erfc(alpha r)/r and its chain (AmoebaReferencePmeMultipoleForce::calculateFixedMultipoleFieldPairIxn :5103-5118);
// cn[n] = (2n - 1)!! / r^(2n+1); lambda_(2n+1), n = 1..4: Thole damping of the rank-n term, lambda_3, lambda_5, lambda_7, lambda_9
// (getDampedInverseDistances :4943-4985; lambda_9 continues the chain: d(lambda_(2n+1) c_n)/dr = -r lambda_(2n+3) c_(n+1)).
// ------------------------------------------------------------------------------------------------
// The damping enters as oml[n] = 1 - lambda_(2n+1), the exponentially small quantity itself (no 1 - (1 - e) in float), and a chain with a
// scale factor s is  bn[n] - ((1 - s) + s oml[n]) cn[n]  (scaled_chain).
template <class T>
__device__ __forceinline__ void pair_chains(T alpha, T r2, T dampI, T dampJ, T tholeI, T tholeJ, T (&bn)[6], T (&cn)[6], T (&oml)[5]) {
    const T r = sqrt(r2), ralpha = alpha * r, invR2 = T(1) / r2;
    const T exp2a = exp(-ralpha * ralpha), alsq2 = T(2) * alpha * alpha;
    T alsq2n = T(1) / (T(MP_SQRT_PI) * alpha);
    bn[0] = erfc(ralpha) / r;
    cn[0] = T(1) / r;
    for (int n = 1; n < 6; n++) {
        alsq2n *= alsq2;
        bn[n] = (T(2 * n - 1) * bn[n - 1] + alsq2n * exp2a) * invR2;
        cn[n] = T(2 * n - 1) * cn[n - 1] * invR2;
    }
    oml[0] = oml[1] = oml[2] = oml[3] = oml[4] = T(0);
    const T damp = dampI * dampJ;
    if (damp != T(0)) {
        const T ratio = r / damp, au3 = (tholeI < tholeJ ? tholeI : tholeJ) * ratio * ratio * ratio;
        if (au3 < T(50)) {
            const T e = exp(-au3);
            oml[1] = e;
            oml[2] = e * (T(1) + au3);
            oml[3] = e * (T(1) + au3 + T(0.6) * au3 * au3);
            oml[4] = e * (T(1) + au3 + (T(18) * au3 * au3 + T(9) * au3 * au3 * au3) / T(35));
        }
    }
}

// MP_SPLIT lanes share an atom in the pair kernels: lane q of the group walks entries q, q + MP_SPLIT, ... of the atom's list and the partial
// sums meet through shuffles.  One thread per atom would put 2 wavefronts on a compute unit at 36 k atoms; the loops are gathers from
// L2 / HBM with long arithmetic in between and need the latency hiding of several wavefronts per SIMD.
#ifndef MP_SPLIT
#define MP_SPLIT 4
#endif
__device__ __forceinline__ double split_sum(double v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);
    if (MP_SPLIT >= 8) v += __shfl_xor(v, 4);
    return v;
}
__device__ __forceinline__ V3 split_sum(V3 v) { return v3(split_sum(v.x), split_sum(v.y), split_sum(v.z)); }

struct PairScale { double m, p, d; };

// scale factors of the pair (i, j) from i's list of special partners (ascending; the cursor moves with j)
// scale factors of a list entry: tag = 1 + index of the partner in this atom's (sorted) row of special pairs, 0 = an ordinary pair
__device__ __forceinline__ PairScale pair_scale(const MpArgs& a, int rowBegin, int entry) {
    PairScale s = {1.0, 1.0, 1.0};
    const int tag = (entry >> 24) & 0x7f;
    if (tag != 0) { const double4 v = a.specScaleSorted[rowBegin + tag - 1]; s.m = v.x; s.p = v.y; s.d = v.z; }
    return s;
}


// ------------------------------------------------------------------------------------------------
// Field of the permanent multipoles at every atom (chains d and p, reciprocal part, self term) and the induced dipoles of direct
// polarization.  calculateFixedMultipoleField (:5169-5201) + calculateFixedMultipoleFieldPairIxn (:5079-5167) + recordFixedMultipoleField
// (:6008-6019) + initializeInducedDipoles (:6021-6026).  Fields in units of e / nm^2 (without the Coulomb constant), as there.
// ------------------------------------------------------------------------------------------------
// MIXED: the pair arithmetic of ordinary pairs in float (separations formed in double, sums kept in double).  The covalently related
// pairs of an atom -- a few, with scale factors that subtract most of the bare kernel at short range: bn - (1 - s) cn loses its digits in
// float there -- are taken from the atom's own row of special partners instead and stay in double: k_mp_special, a launch of its own
// ahead of this one (inside this kernel its registers -- 256 and more for the force variant -- would set the occupancy of the float loop).
template <bool MIXED>
__global__ __launch_bounds__(MP_BLOCK) void k_mp_field(MpArgs a) {
    const int t = threadIdx.x, g = (blockIdx.x * MP_BLOCK + t) / MP_SPLIT, q = t % MP_SPLIT, i = scan_atom(a, g);
    const bool active = i >= 0;
    const int ii = active ? i : 0;
    const V3 xi = position(a, ii);
    const double tholeI = a.thole[ii], dampI = a.damping[ii];
    const int rowBegin = a.specStart[ii];
    V3 ed = v3(0, 0, 0), ep = v3(0, 0, 0);
    {
        PlSpan span = {0, 0, 0, 0};
        if (active) span = pl_span(a.pairCount, a.listStride, g);
        for (int k = q; k < span.total; k += MP_SPLIT) {
            const int entry = pl_at(a.pairList, a.listStride, a.listSubcap, span, k, g);
            const int j = scan_atom(a, entry & PL_POS_MASK);
            const bool tagged = ((entry >> 24) & 0x7f) != 0;
            const V3 xj = position(a, j);
            double dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
            min_image_d(dx, dy, dz, a.box);
            const double r2 = dx * dx + dy * dy + dz * dz;
            const bool inside = !(r2 > a.cutoff2);
            const size_t plane = (size_t) a.pairCap * a.listStride, at = (size_t) k * a.listStride + g;
            if (a.pairCache != nullptr) { a.pairCache[at] = (float) dx; a.pairCache[plane + at] = (float) dy; a.pairCache[2 * plane + at] = (float) dz; }
            if (!inside) {
                if (a.pairCache != nullptr) { a.pairCache[3 * plane + at] = 0.f; a.pairCache[4 * plane + at] = 0.f; }
                continue;
            }
            Site Mj;
            Mj.q = a.charge[j]; Mj.mu = load3(a.labDipole, j); Mj.Q = load6(a.labQuad, j);
            if (MIXED) {
                float bn[6], cn[6], oml[5];
                pair_chains<float>((float) a.alpha, (float) r2, (float) dampI, (float) a.damping[j], (float) tholeI, (float) a.thole[j], bn, cn, oml);
                // what the induced-dipole field of every solver iteration needs of this pair (k_mp_dipole_field): geometry and the two
                // coefficients of the damped chain -- the erfc / exp / Thole arithmetic is done once per evaluation, not once per iteration.
                // Stored as float: the iterations stream this cache, their sums stay double, and the converged dipoles only have to meet the
                // solver's tolerance; the forces do not read it.
                if (a.pairCache != nullptr) { a.pairCache[3 * plane + at] = bn[1] - oml[1] * cn[1]; a.pairCache[4 * plane + at] = bn[2] - oml[2] * cn[2]; }
                if (tagged) continue;                      // second loop, in double
                V3T<float> fd, fp;
                field_pair<float>(convert_site<float>(Mj), v3t<float>((float) dx, (float) dy, (float) dz), bn, cn, oml, 1.f, 1.f, fd, fp);
                ed = ed + widen3(fd); ep = ep + widen3(fp);
            }
            else {
                double bn[6], cn[6], oml[5];
                pair_chains<double>(a.alpha, r2, dampI, a.damping[j], tholeI, a.thole[j], bn, cn, oml);
                if (a.pairCache != nullptr) { a.pairCache[3 * plane + at] = (float) (bn[1] - oml[1] * cn[1]); a.pairCache[4 * plane + at] = (float) (bn[2] - oml[2] * cn[2]); }
                const PairScale sc = pair_scale(a, rowBegin, entry);
                V3 fd, fp;
                field_pair<double>(Mj, v3(dx, dy, dz), bn, cn, oml, sc.d, sc.p, fd, fp);
                ed = ed + fd; ep = ep + fp;
            }
        }
    }
    ed = split_sum(ed); ep = split_sum(ep);
    if (!active || q != 0) return;
    if (MIXED) { ed = ed + load3(a.fieldD, i); ep = ep + load3(a.fieldP, i); }       // the covalently related partners: k_mp_special<false>, launched before
    if (a.pairsOnly) {
        // the reciprocal potential is still being formed on the side stream: k_mp_field_finish adds it (and the self field) behind the wait
        store3(a.fieldD, i, ed); store3(a.fieldP, i, ep);
        return;
    }
    // reciprocal field -grad phi (the grid carries the Coulomb constant: taken out again) and the self field 4 alpha^3 / (3 sqrt(pi)) mu
    const double* phi = a.phi + 20 * (size_t) i;
    const double selfTerm = (4.0 / 3.0) * a.alpha * a.alpha * a.alpha / MP_SQRT_PI;
    const V3 common = v3(-phi[1], -phi[2], -phi[3]) * (1.0 / OMM_ONE_4PI_EPS0_D) + selfTerm * load3(a.labDipole, i);
    ed = ed + common; ep = ep + common;
    store3(a.fieldD, i, ed); store3(a.fieldP, i, ep);
    const double pol = a.polarity[i];
    store3(a.indD, i, pol * ed); store3(a.indP, i, pol * ep);
}

// the last lines of k_mp_field as a launch of their own (MpArgs::pairsOnly): reciprocal and self field added to the pair sums, direct dipoles
__global__ void k_mp_field_finish(MpArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const double* phi = a.phi + 20 * (size_t) i;
    const double selfTerm = (4.0 / 3.0) * a.alpha * a.alpha * a.alpha / MP_SQRT_PI;
    const V3 common = v3(-phi[1], -phi[2], -phi[3]) * (1.0 / OMM_ONE_4PI_EPS0_D) + selfTerm * load3(a.labDipole, i);
    const V3 ed = load3(a.fieldD, i) + common, ep = load3(a.fieldP, i) + common;
    store3(a.fieldD, i, ed); store3(a.fieldP, i, ep);
    const double pol = a.polarity[i];
    store3(a.indD, i, pol * ed); store3(a.indP, i, pol * ep);
}

// ------------------------------------------------------------------------------------------------
// Energy, forces and torques: real-space pairs (permanent x permanent through chain m, permanent x induced through chains p and d),
// reciprocal space from the two sets of potential derivatives, self terms.  calculatePmeDirectElectrostaticPairIxn (:6335-6751, here in
// the Cartesian form described at the top), computeReciprocalSpaceFixedMultipoleForceAndEnergy (:5820-5897),
// computeReciprocalSpaceInducedDipoleForceAndEnergy (:5899-6006), calculatePmeSelfEnergy / calculatePmeSelfTorque (:6294-6333).
// ------------------------------------------------------------------------------------------------
// energy, force and torque of a multipole with moments (q, mu, Q) in a potential given by its derivatives p[0..19]
__device__ __forceinline__ void in_potential(double q, V3 mu, const Sym& Q, const double* p, double& energy, V3& force, V3& torque) {
    const V3 g = v3(p[1], p[2], p[3]);
    const Sym H = {p[4], p[5], p[6], p[7], p[8], p[9]};
    energy = q * p[0] + dot(mu, g) + ddot(Q, H);
    // third derivatives: xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz
    const double xxx = p[10], xxy = p[11], xxz = p[12], xyy = p[13], xyz = p[14], xzz = p[15], yyy = p[16], yyz = p[17], yzz = p[18], zzz = p[19];
    const V3 QD = v3(Q.xx * xxx + Q.yy * xyy + Q.zz * xzz + 2.0 * (Q.xy * xxy + Q.xz * xxz + Q.yz * xyz),
                     Q.xx * xxy + Q.yy * yyy + Q.zz * yzz + 2.0 * (Q.xy * xyy + Q.xz * xyz + Q.yz * yyz),
                     Q.xx * xxz + Q.yy * yyz + Q.zz * zzz + 2.0 * (Q.xy * xyz + Q.xz * xzz + Q.yz * yzz));
    force = v3(0, 0, 0) - (q * g + mul(H, mu) + QD);
    torque = v3(0, 0, 0) - cross(mu, g) - 2.0 * asym_product(Q, H);
}

// Reciprocal-space and self terms of atom i (the potentials carry the Coulomb constant already): added to energy, force and torque.
// computeReciprocalSpaceFixedMultipoleForceAndEnergy (:5820-5897), computeReciprocalSpaceInducedDipoleForceAndEnergy (:5899-6006),
// calculatePmeSelfEnergy / calculatePmeSelfTorque (:6294-6333).
__device__ __forceinline__ void reciprocal_and_self(const MpArgs& a, int i, const Site& Mi, V3 udI, V3 upI, double& energy, V3& force, V3& torque) {
    const V3 nu = 0.5 * (udI + upI);
    // ---- reciprocal space (the potentials carry the Coulomb constant already)
    const double* phi = a.phi + 20 * (size_t) i;
    double phiNu[20];           // potential of (mu_d + mu_p) / 2
    for (int k = 0; k < 20; k++) phiNu[k] = a.mutual ? 0.5 * (a.phiInd[20 * (size_t) i + k] + a.phiIndP[20 * (size_t) i + k]) : a.phiInd[20 * (size_t) i + k];
    const double* phiInd = phiNu;
    double e; V3 f, tq;
    in_potential(Mi.q, Mi.mu, Mi.Q, phi, e, f, tq);            // permanent multipoles in the potential of all permanent multipoles
    energy += 0.5 * e; force = force + f; torque = torque + tq;
    in_potential(Mi.q, Mi.mu, Mi.Q, phiInd, e, f, tq);         // ... and in the potential of the induced dipoles (mu_d + mu_p) / 2
    force = force + f; torque = torque + tq;
    const Sym zero = {0, 0, 0, 0, 0, 0};
    in_potential(0.0, nu, zero, phi, e, f, tq);                // the induced dipole in the potential of the permanent multipoles
    energy += 0.5 * e; force = force + f;
    if (a.mutual) {
        // -1/2 [mu_d . grad grad phi(mu_p) + mu_p . grad grad phi(mu_d)]  (computeReciprocalSpaceInducedDipoleForceAndEnergy :5976-5980)
        in_potential(0.0, udI, zero, a.phiIndP + 20 * (size_t) i, e, f, tq);
        force = force + 0.5 * f;
        in_potential(0.0, upI, zero, a.phiInd + 20 * (size_t) i, e, f, tq);
        force = force + 0.5 * f;
    }
    // ---- self terms
    const double a2 = a.alpha * a.alpha, prefac = -a.alpha * OMM_ONE_4PI_EPS0_D / MP_SQRT_PI;
    const double dxy = Mi.Q.xx - Mi.Q.yy;
    const double qii = 9.0 * (Mi.Q.zz * Mi.Q.zz + (4.0 / 3.0) * (Mi.Q.xz * Mi.Q.xz + Mi.Q.yz * Mi.Q.yz + Mi.Q.xy * Mi.Q.xy) + (1.0 / 3.0) * dxy * dxy);
    energy += prefac * (Mi.q * Mi.q + (2.0 / 3.0) * a2 * dot(Mi.mu, Mi.mu + nu) + (4.0 / 15.0) * a2 * a2 * qii);
    torque = torque + ((4.0 / 3.0) * OMM_ONE_4PI_EPS0_D * a2 * a.alpha / MP_SQRT_PI) * cross(Mi.mu, nu);
}

// The covalently related partners of every atom (its row of special pairs: a handful, scale factors m / p / d below one) in double, for
// the mixed-precision kernels: FORCES = false -- their part of the fixed field into fieldD / fieldP (k_mp_field<true> adds the rest);
// FORCES = true -- their energy and force and the atom's reciprocal-space and self terms (added here), the torque of both left in a.torque
// (k_mp_forces<true> adds the list pairs).  MP_SPLIT lanes per atom, as in the list kernels.
template <bool FORCES>
__global__ __launch_bounds__(MP_BLOCK) void k_mp_special(MpArgs a) {
    __shared__ double sEnergy[MP_BLOCK / 64];
    const int t = threadIdx.x, i = (blockIdx.x * MP_BLOCK + t) / MP_SPLIT, q = t % MP_SPLIT;
    const bool active = i < a.n;
    const int ii = active ? i : 0;
    const V3 xi = position(a, ii);
    Site Mi;
    Mi.q = a.charge[ii]; Mi.mu = load3(a.labDipole, ii); Mi.Q = load6(a.labQuad, ii);
    const V3 udI = FORCES ? load3(a.indD, ii) : v3(0, 0, 0), upI = FORCES ? load3(a.indP, ii) : v3(0, 0, 0);
    const double tholeI = a.thole[ii], dampI = a.damping[ii];
    const int rowBegin = a.specStart[ii], rowEnd = active ? a.specStart[ii + 1] : rowBegin;
    V3 accA = v3(0, 0, 0), accB = v3(0, 0, 0);          // field d / p, or force / torque
    double energy = 0.0;
    for (int c = rowBegin + q; c < rowEnd; c += MP_SPLIT) {
        const int j = a.specAtom[c];
        const double4 scale = a.specScale[c];              // (m, p, d, u)
        const V3 xj = position(a, j);
        double dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
        min_image_d(dx, dy, dz, a.box);
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 > a.cutoff2) continue;
        Site Mj;
        Mj.q = a.charge[j]; Mj.mu = load3(a.labDipole, j); Mj.Q = load6(a.labQuad, j);
        double bn[6], cn[6], oml[5];
        pair_chains<double>(a.alpha, r2, dampI, a.damping[j], tholeI, a.thole[j], bn, cn, oml);
        if (FORCES) {
            double e; V3 f, tq;
            forces_pair<double>(Mi, Mj, udI, upI, load3(a.indD, j), load3(a.indP, j), v3(dx, dy, dz), bn, cn, oml, scale.x, scale.y, scale.z, a.mutual != 0, e, f, tq);
            energy += e; accA = accA + f; accB = accB + tq;
        }
        else {
            V3 fd, fp;
            field_pair<double>(Mj, v3(dx, dy, dz), bn, cn, oml, scale.z, scale.y, fd, fp);
            accA = accA + fd; accB = accB + fp;
        }
    }
    accA = split_sum(accA); accB = split_sum(accB); energy = split_sum(energy);
    if (q != 0) energy = 0.0;
    if (active && q == 0) {
        if (FORCES) {
            accA = OMM_ONE_4PI_EPS0_D * accA; accB = OMM_ONE_4PI_EPS0_D * accB; energy *= OMM_ONE_4PI_EPS0_D;
            reciprocal_and_self(a, i, Mi, udI, upI, energy, accA, accB);           // everything of the atom that is not a pair of the list
            if (a.torque2 != nullptr) store3(a.torque2, i, accB);
            else store3(a.torque, i, a.specialAdds ? accB + load3(a.torque, i) : accB);
            add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], accA.x, accA.y, accA.z);
        }
        else { store3(a.fieldD, i, accA); store3(a.fieldP, i, accB); }
    }
    if (FORCES && a.includeEnergy) {
        energy = wave_sum(active ? energy : 0.0);
        if ((t & 63) == 0) sEnergy[t >> 6] = energy;
        __syncthreads();
        if (t == 0) {
            double e = 0;
            for (int w = 0; w < MP_BLOCK / 64; w++) e += sEnergy[w];
            atomicAdd(&a.energyBuffer[blockIdx.x % a.energySlots], e);
        }
    }
}

template <bool MIXED>
__global__ __launch_bounds__(MP_BLOCK) void k_mp_forces(MpArgs a) {
    __shared__ double sEnergy[MP_BLOCK / 64];
    const int t = threadIdx.x, g = (blockIdx.x * MP_BLOCK + t) / MP_SPLIT, q = t % MP_SPLIT, i = scan_atom(a, g);
    const bool active = i >= 0;
    const int ii = active ? i : 0;
    const V3 xi = position(a, ii);
    Site Mi;
    Mi.q = a.charge[ii]; Mi.mu = load3(a.labDipole, ii); Mi.Q = load6(a.labQuad, ii);
    const V3 udI = load3(a.indD, ii), upI = load3(a.indP, ii);
    const double tholeI = a.thole[ii], dampI = a.damping[ii];
    const int rowBegin = a.specStart[ii];
    V3 force = v3(0, 0, 0), torque = v3(0, 0, 0);
    double energy = 0.0;
    {
        // (MIXED: ordinary pairs in float from the list; the atom's covalently related partners in double by k_mp_special<true>: see k_mp_field)
        const SiteT<float> MiF = convert_site<float>(Mi);
        const V3T<float> udIF = convert3<float>(udI), upIF = convert3<float>(upI);
        PlSpan span = {0, 0, 0, 0};
        if (active) span = pl_span(a.pairCount, a.listStride, g);
        for (int k = q; k < span.total; k += MP_SPLIT) {
            const int entry = pl_at(a.pairList, a.listStride, a.listSubcap, span, k, g);
            if (MIXED && ((entry >> 24) & 0x7f) != 0) continue;
            const int j = scan_atom(a, entry & PL_POS_MASK);
            const V3 xj = position(a, j);
            double dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
            min_image_d(dx, dy, dz, a.box);
            const double r2 = dx * dx + dy * dy + dz * dz;
            if (r2 > a.cutoff2) continue;
            Site Mj;
            Mj.q = a.charge[j]; Mj.mu = load3(a.labDipole, j); Mj.Q = load6(a.labQuad, j);
            const V3 udJ = load3(a.indD, j), upJ = load3(a.indP, j);
            if (MIXED) {
                float bn[6], cn[6], oml[5], e;
                pair_chains<float>((float) a.alpha, (float) r2, (float) dampI, (float) a.damping[j], (float) tholeI, (float) a.thole[j], bn, cn, oml);
                V3T<float> f, tq;
                forces_pair<float>(MiF, convert_site<float>(Mj), udIF, upIF, convert3<float>(udJ), convert3<float>(upJ), v3t<float>((float) dx, (float) dy, (float) dz),
                                   bn, cn, oml, 1.f, 1.f, 1.f, a.mutual != 0, e, f, tq);
                energy += (double) e; force = force + widen3(f); torque = torque + widen3(tq);
            }
            else {
                const PairScale sc = pair_scale(a, rowBegin, entry);
                double bn[6], cn[6], oml[5], e;
                pair_chains<double>(a.alpha, r2, dampI, a.damping[j], tholeI, a.thole[j], bn, cn, oml);
                V3 f, tq;
                forces_pair<double>(Mi, Mj, udI, upI, udJ, upJ, v3(dx, dy, dz), bn, cn, oml, sc.m, sc.p, sc.d, a.mutual != 0, e, f, tq);
                energy += e; force = force + f; torque = torque + tq;
            }
        }
    }
    force = split_sum(force); torque = split_sum(torque); energy = split_sum(energy);
    if (q != 0) energy = 0.0;
    if (active && q == 0) {
        // pair quantities carry the Coulomb constant from here on
        force = OMM_ONE_4PI_EPS0_D * force; torque = OMM_ONE_4PI_EPS0_D * torque; energy *= OMM_ONE_4PI_EPS0_D;
        if (!MIXED) reciprocal_and_self(a, i, Mi, udI, upI, energy, force, torque);          // (MIXED: k_mp_special<true> has added them)
        if (MIXED && !a.specialAdds) torque = torque + load3(a.torque, i);          // the covalently related partners: k_mp_special<true>, launched before (Coulomb constant included) -- or behind, adding to these (specialAdds)
        store3(a.torque, i, torque);
        add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], force.x, force.y, force.z);
    }
    if (a.includeEnergy) {
        energy = wave_sum(active ? energy : 0.0);
        if ((t & 63) == 0) sEnergy[t >> 6] = energy;
        __syncthreads();
        if (t == 0) {
            double e = 0;
            for (int w = 0; w < MP_BLOCK / 64; w++) e += sEnergy[w];
            atomicAdd(&a.energyBuffer[blockIdx.x % a.energySlots], e);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Mutual polarization: field of two sets of dipoles (vD, vP) at every atom -- real space through the Thole-damped chain, the
// reciprocal part from their two potentials, the self field.  calculateInducedDipoleFields (:6059-6152).
// ------------------------------------------------------------------------------------------------

// cgStage >= 0 (solver work vectors in w, no preconditioner): the per-atom part of the conjugate-gradient stage that consumes this field is
// done here instead of in a launch of its own (k_mp_cg describes the stages) -- 0: the residual of the first guess, z = p = alpha r and
// their dot products; 1: A p = p / alpha - T p into the place of T p, and p . A p.  -1: the field is stored, nothing else.
__global__ __launch_bounds__(MP_BLOCK) void k_mp_dipole_field(MpArgs a, const double* __restrict__ vD, const double* __restrict__ vP, const double* __restrict__ phiD,
                                                              const double* __restrict__ phiP, double* __restrict__ outD, double* __restrict__ outP, double* w, int cgStage) {
    if (a.doneFlag != nullptr && (a.doneFlag[0] != 0.0 || a.doneFlag[3] != 0.0)) return;           // an iteration enqueued ahead of the convergence check (solve_mutual)
    const int t = threadIdx.x, g = (blockIdx.x * MP_BLOCK + t) / MP_SPLIT, q = t % MP_SPLIT, i = scan_atom(a, g);
    const bool active = i >= 0;
    const int ii = active ? i : 0;
    const V3 xi = position(a, ii);
    const double tholeI = a.thole[ii], dampI = a.damping[ii];
    V3 ed = v3(0, 0, 0), ep = v3(0, 0, 0);
    {
        PlSpan span = {0, 0, 0, 0};
        if (active) span = pl_span(a.pairCount, a.listStride, g);
        // (four entries in flight per lane, `#pragma unroll 4`: measured 1-2 % slower on both AMOEBA workloads, profiles/r11/r11l_*)
        for (int k = q; k < span.total; k += MP_SPLIT) {
            const int sj = pl_at(a.pairList, a.listStride, a.listSubcap, span, k, g) & PL_POS_MASK;
            struct { V3 vd, vp; } s;
            int j = -1;
            if (a.gather != nullptr) {
                // the partner's dipoles from the packed copy in scan order: no slot -> atom indirection, 24 bytes instead of 48, and partners
                // that follow each other in the list (spatial neighbours) share cache lines -- the atom-ordered doubles are scattered
                const float* v = a.gather + 6 * (size_t) sj;
                s.vd = v3(v[0], v[1], v[2]); s.vp = v3(v[3], v[4], v[5]);
            }
            else { j = scan_atom(a, sj); s.vd = load3(vD, j); s.vp = load3(vP, j); }
            V3 r;
            double b1, b2;
            if (a.pairCache != nullptr) {
                const size_t plane = (size_t) a.pairCap * a.listStride, at = (size_t) k * a.listStride + g;
                r = v3(a.pairCache[at], a.pairCache[plane + at], a.pairCache[2 * plane + at]);
                b1 = a.pairCache[3 * plane + at]; b2 = a.pairCache[4 * plane + at];
            }
            else {
                if (j < 0) j = scan_atom(a, sj);
                const V3 xj = position(a, j);
                double dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
                min_image_d(dx, dy, dz, a.box);
                const double r2 = dx * dx + dy * dy + dz * dz;
                if (r2 > a.cutoff2) continue;
                double bn[6], cn[6], oml[5];
                pair_chains<double>(a.alpha, r2, dampI, a.damping[j], tholeI, a.thole[j], bn, cn, oml);
                r = v3(dx, dy, dz);
                b1 = bn[1] - oml[1] * cn[1]; b2 = bn[2] - oml[2] * cn[2];
            }
            ed = ed + (b2 * dot(s.vd, r)) * r - b1 * s.vd;
            ep = ep + (b2 * dot(s.vp, r)) * r - b1 * s.vp;
        }
    }
    ed = split_sum(ed); ep = split_sum(ep);
    if (cgStage == 2) {
        // pairs only: the real-space part, stored raw -- the reciprocal part is being formed on the side stream meanwhile, k_mp_cg stage 9 (or 10)
        // puts the two together (solve_mutual, overlap mode)
        if (active && q == 0) { store3(outD, i, ed); store3(outP, i, ep); }
        return;
    }
    if (cgStage < 0 && (!active || q != 0)) return;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (active && q == 0) {
        const double selfTerm = (4.0 / 3.0) * a.alpha * a.alpha * a.alpha / MP_SQRT_PI, invK = 1.0 / OMM_ONE_4PI_EPS0_D;
        const double* pd = phiD + 20 * (size_t) i;
        const double* pp = phiP + 20 * (size_t) i;
        const V3 viD = load3(vD, i), viP = load3(vP, i);
        const V3 td = ed - invK * v3(pd[1], pd[2], pd[3]) + selfTerm * viD, tp = ep - invK * v3(pp[1], pp[2], pp[3]) + selfTerm * viP;
        if (cgStage < 0) { store3(outD, i, td); store3(outP, i, tp); }
        else {
            const size_t n3 = 3 * (size_t) a.n;
            const double pol = a.polarity[i], invPol = pol > 0 ? 1.0 / pol : 0.0;
            if (cgStage == 0) {
                V3 rd = v3(0, 0, 0), rp = v3(0, 0, 0);
                if (pol > 0) { rd = load3(a.fieldD, i) - invPol * viD + td; rp = load3(a.fieldP, i) - invPol * viP + tp; }
                store3(w, i, rd); store3(w + n3, i, rp);
                store3(w + 2 * n3, i, pol * rd); store3(w + 3 * n3, i, pol * rp); store3(w + 4 * n3, i, pol * rd); store3(w + 5 * n3, i, pol * rp);
                // (the packed copy of p for the first iteration is written by k_mp_pack after this launch: other lanes are still reading the copy of mu_0)
                s0 = pol * dot(rd, rd); s1 = pol * dot(rp, rp); s2 = pol * pol * dot(rd, rd); s3 = pol * pol * dot(rp, rp);
            }
            else {
                const V3 ad = invPol * viD - (pol > 0 ? td : v3(0, 0, 0)), ap = invPol * viP - (pol > 0 ? tp : v3(0, 0, 0));
                store3(outD, i, ad); store3(outP, i, ap);            // t now holds A p
                s0 = dot(viD, ad); s1 = dot(viP, ap);
            }
        }
    }
    if (cgStage < 0) return;
    double* sums = w + 8 * 3 * (size_t) a.n;
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if (cgStage == 0) { s2 = wave_sum(s2); s3 = wave_sum(s3); }
    if ((t & 63) == 0) {
        if (cgStage == 0) { atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); atomicAdd(&sums[4], s2); atomicAdd(&sums[5], s3); }
        else { atomicAdd(&sums[2], s0); atomicAdd(&sums[3], s1); }
    }
}

// ------------------------------------------------------------------------------------------------
// Extrapolated polarization (OPT; Simmonett, Pickard, Ponder, Brooks, J. Chem. Phys. 145, 164101 (2016);
// AmoebaReferenceMultipoleForce::convergeInduceDipolesByExtrapolation :891-937): mu_0 = alpha E, mu_(n+1) = alpha T mu_n, the induced
// dipoles are sum_n P_n mu_n with P_n the partial sums of the extrapolation coefficients.  No iteration to convergence: K - 1 field
// evaluations.  Energy and forces are those of direct polarization evaluated with the total dipoles plus, per atom,
//     F_i += 1/2 sum_(l + m + 1 < K) P_(l+m+1) [ mu_d^(l) . G_p^(m) + mu_p^(l) . G_d^(m) ]           (:6787-6816)
// where G^(m) is the GRADIENT at atom i of the field that the dipoles mu^(m) produce -- so every field evaluation also returns that:
// real space from the Thole-damped chain one order up (calculateDirectInducedDipolePairIxns :6172-6290: E_ab = (mu.r) r_a r_b b3
// - (mu_a r_b + mu_b r_a + delta_ab mu.r) b2), reciprocal space from the second derivatives of the dipoles' potential (:6080-6140).
// ------------------------------------------------------------------------------------------------
// field (3) and field gradient (xx, yy, zz, xy, xz, yz) of the dipole sets vD, vP at every atom.  MIXED: the pair chains in float.
template <bool MIXED>
__global__ __launch_bounds__(MP_BLOCK) void k_mp_dipole_field_gradient(MpArgs a, const double* __restrict__ vD, const double* __restrict__ vP, const double* __restrict__ phiD,
                                                                       const double* __restrict__ phiP, double* __restrict__ outD, double* __restrict__ outP,
                                                                       double* __restrict__ gradD, double* __restrict__ gradP) {
    const int t = threadIdx.x, g = (blockIdx.x * MP_BLOCK + t) / MP_SPLIT, q = t % MP_SPLIT, i = scan_atom(a, g);
    const bool active = i >= 0;
    const int ii = active ? i : 0;
    const V3 xi = position(a, ii);
    const double tholeI = a.thole[ii], dampI = a.damping[ii];
    V3 ed = v3(0, 0, 0), ep = v3(0, 0, 0);
    double gd[6] = {0, 0, 0, 0, 0, 0}, gp[6] = {0, 0, 0, 0, 0, 0};
    PlSpan span = {0, 0, 0, 0};
    if (active) span = pl_span(a.pairCount, a.listStride, g);
    for (int k = q; k < span.total; k += MP_SPLIT) {
        const int j = scan_atom(a, pl_at(a.pairList, a.listStride, a.listSubcap, span, k, g) & PL_POS_MASK);
        const V3 xj = position(a, j);
        double dx = xj.x - xi.x, dy = xj.y - xi.y, dz = xj.z - xi.z;
        min_image_d(dx, dy, dz, a.box);
        const double r2 = dx * dx + dy * dy + dz * dz;
        if (r2 > a.cutoff2) continue;
        double b1, b2, b3;
        if (MIXED) {
            float bn[6], cn[6], oml[5];
            pair_chains<float>((float) a.alpha, (float) r2, (float) dampI, (float) a.damping[j], (float) tholeI, (float) a.thole[j], bn, cn, oml);
            b1 = bn[1] - oml[1] * cn[1]; b2 = bn[2] - oml[2] * cn[2]; b3 = bn[3] - oml[3] * cn[3];
        }
        else {
            double bn[6], cn[6], oml[5];
            pair_chains<double>(a.alpha, r2, dampI, a.damping[j], tholeI, a.thole[j], bn, cn, oml);
            b1 = bn[1] - oml[1] * cn[1]; b2 = bn[2] - oml[2] * cn[2]; b3 = bn[3] - oml[3] * cn[3];
        }
        const V3 r = v3(dx, dy, dz);
        const V3 vd = load3(vD, j), vp = load3(vP, j);
        const double md = dot(vd, r), mp = dot(vp, r);
        ed = ed + (b2 * md) * r - b1 * vd;
        ep = ep + (b2 * mp) * r - b1 * vp;
        gd[0] += md * dx * dx * b3 - (2.0 * vd.x * dx + md) * b2; gd[1] += md * dy * dy * b3 - (2.0 * vd.y * dy + md) * b2; gd[2] += md * dz * dz * b3 - (2.0 * vd.z * dz + md) * b2;
        gd[3] += md * dx * dy * b3 - (vd.x * dy + vd.y * dx) * b2; gd[4] += md * dx * dz * b3 - (vd.x * dz + vd.z * dx) * b2; gd[5] += md * dy * dz * b3 - (vd.y * dz + vd.z * dy) * b2;
        gp[0] += mp * dx * dx * b3 - (2.0 * vp.x * dx + mp) * b2; gp[1] += mp * dy * dy * b3 - (2.0 * vp.y * dy + mp) * b2; gp[2] += mp * dz * dz * b3 - (2.0 * vp.z * dz + mp) * b2;
        gp[3] += mp * dx * dy * b3 - (vp.x * dy + vp.y * dx) * b2; gp[4] += mp * dx * dz * b3 - (vp.x * dz + vp.z * dx) * b2; gp[5] += mp * dy * dz * b3 - (vp.y * dz + vp.z * dy) * b2;
    }
    ed = split_sum(ed); ep = split_sum(ep);
    for (int c = 0; c < 6; c++) { gd[c] = split_sum(gd[c]); gp[c] = split_sum(gp[c]); }
    if (!active || q != 0) return;
    const double selfTerm = (4.0 / 3.0) * a.alpha * a.alpha * a.alpha / MP_SQRT_PI, invK = 1.0 / OMM_ONE_4PI_EPS0_D;
    const double* pd = phiD + 20 * (size_t) i;
    const double* pp = phiP + 20 * (size_t) i;
    store3(outD, i, ed - invK * v3(pd[1], pd[2], pd[3]) + selfTerm * load3(vD, i));
    store3(outP, i, ep - invK * v3(pp[1], pp[2], pp[3]) + selfTerm * load3(vP, i));
    // reciprocal part of the gradient: minus the second derivatives of the potential, out[4..9] = xx xy xz yy yz zz
    const int at[6] = {4, 7, 9, 5, 6, 8};
    for (int c = 0; c < 6; c++) { gradD[6 * (size_t) i + c] = gd[c] - invK * pd[at[c]]; gradP[6 * (size_t) i + c] = gp[c] - invK * pp[at[c]]; }
}

// stage 0: record the dipoles in indD / indP as order `order`;  stage 1: the next order, alpha x the field in fD / fP, into indD / indP and
// the record;  stage 2: indD / indP = sum_n P_n record_n
struct ExtCoeff { double p[OMMHIP_AMOEBA_MAX_EXT_ORDERS]; };
__global__ void k_mp_ext_step(MpArgs a, double* records, int order, int orders, int stage, const double* __restrict__ fD, const double* __restrict__ fP, ExtCoeff coeff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const size_t n3 = 3 * (size_t) a.n;
    if (stage == 2) {
        V3 d = v3(0, 0, 0), p = v3(0, 0, 0);
        for (int k = 0; k < orders; k++) { d = d + coeff.p[k] * load3(records + (size_t) k * 2 * n3, i); p = p + coeff.p[k] * load3(records + (size_t) k * 2 * n3 + n3, i); }
        store3(a.indD, i, d); store3(a.indP, i, p);
        return;
    }
    if (stage == 1) { const double pol = a.polarity[i]; store3(a.indD, i, pol * load3(fD, i)); store3(a.indP, i, pol * load3(fP, i)); }
    double* rec = records + (size_t) order * 2 * n3;
    store3(rec, i, load3(a.indD, i)); store3(rec + n3, i, load3(a.indP, i));
}

// the dipole-response part of the extrapolated-polarization force (see above); gradients: `orders - 1` records of (G_d, G_p), 12 n doubles each
__global__ void k_mp_ext_forces(MpArgs a, const double* __restrict__ records, const double* __restrict__ gradients, int orders, ExtCoeff coeff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const size_t n3 = 3 * (size_t) a.n, n6 = 6 * (size_t) a.n;
    V3 f = v3(0, 0, 0);
    for (int l = 0; l < orders - 1; l++)
        for (int m = 0; m < orders - 1 - l; m++) {
            const double p = coeff.p[l + m + 1];
            if (fabs(p) < 1e-6) continue;
            const V3 ud = load3(records + (size_t) l * 2 * n3, i), up = load3(records + (size_t) l * 2 * n3 + n3, i);
            const double* gD = gradients + (size_t) m * 2 * n6 + 6 * (size_t) i;
            const double* gP = gradients + (size_t) m * 2 * n6 + n6 + 6 * (size_t) i;
            // (xx, yy, zz, xy, xz, yz)
            f = f + (0.5 * p) * v3(ud.x * gP[0] + ud.y * gP[3] + ud.z * gP[4], ud.x * gP[3] + ud.y * gP[1] + ud.z * gP[5], ud.x * gP[4] + ud.y * gP[5] + ud.z * gP[2]);
            f = f + (0.5 * p) * v3(up.x * gD[0] + up.y * gD[3] + up.z * gD[4], up.x * gD[3] + up.y * gD[1] + up.z * gD[5], up.x * gD[4] + up.y * gD[5] + up.z * gD[2]);
        }
    f = OMM_ONE_4PI_EPS0_D * f;
    add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], f.x, f.y, f.z);
}

// Preconditioner of the conjugate gradients: z = M r with M = 2 alpha + alpha T_near alpha, the first terms of the Neumann series of
// (1/alpha - T)^-1 with the pair tensor kept for partners closer than 0.45 nm and the diagonal doubled -- Tinker's choice (induce.f, uscale0a /
// uscale0b: udiag = 2, usolvcut = 4.5 A).  Measured here: 8 -> 7 iterations on 12 167 waters at epsilon 1e-5, which does not pay for the extra list walk: opt-in.  The pairs and their damped tensor coefficients come from
// the pair cache k_mp_field wrote; symmetric by construction (T_ij = T_ji, both directions visited).  Also accumulates r.z: into sums[0,1] at
// the start (initial = 1: p = z as well), into sums[6,7] inside an iteration.
__global__ __launch_bounds__(MP_BLOCK) void k_mp_precond(MpArgs a, double* w, int initial) {
    if (a.doneFlag != nullptr && (a.doneFlag[0] != 0.0 || a.doneFlag[3] != 0.0)) return;
    const int t = threadIdx.x, g = (blockIdx.x * MP_BLOCK + t) / MP_SPLIT, q = t % MP_SPLIT, i = scan_atom(a, g);
    const bool active = i >= 0;
    const size_t n3 = 3 * (size_t) a.n;
    double* sums = w + 8 * n3;
    const double* rD = w; const double* rP = w + n3;
    double* zD = w + 2 * n3; double* zP = w + 3 * n3; double* pD = w + 4 * n3; double* pP = w + 5 * n3;
    V3 zd = v3(0, 0, 0), zp = v3(0, 0, 0);
    if (active) {
        const PlSpan span = pl_span(a.pairCount, a.listStride, g);
        const size_t plane = (size_t) a.pairCap * a.listStride;
        for (int k = q; k < span.total; k += MP_SPLIT) {
            const size_t at = (size_t) k * a.listStride + g;
            const V3 r = v3(a.pairCache[at], a.pairCache[plane + at], a.pairCache[2 * plane + at]);
            if (dot(r, r) > a.precondCut2) continue;
            const int j = scan_atom(a, pl_at(a.pairList, a.listStride, a.listSubcap, span, k, g) & PL_POS_MASK);
            const double b1 = a.pairCache[3 * plane + at], b2 = a.pairCache[4 * plane + at], polJ = a.polarity[j];
            const V3 vd = load3(rD, j), vp = load3(rP, j);
            zd = zd + polJ * ((b2 * dot(vd, r)) * r - b1 * vd);
            zp = zp + polJ * ((b2 * dot(vp, r)) * r - b1 * vp);
        }
    }
    zd = split_sum(zd); zp = split_sum(zp);
    double s0 = 0, s1 = 0;
    if (active && q == 0) {
        const double pol = a.polarity[i];
        const V3 rd = load3(rD, i), rp = load3(rP, i);
        zd = pol * (2.0 * rd + zd); zp = pol * (2.0 * rp + zp);
        store3(zD, i, zd); store3(zP, i, zp);
        if (initial) { store3(pD, i, zd); store3(pP, i, zp); }
        s0 = dot(rd, zd); s1 = dot(rp, zp);
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) { atomicAdd(&sums[initial ? 0 : 6], s0); atomicAdd(&sums[initial ? 1 : 7], s1); }
}

// Conjugate gradients on (1/alpha - T) mu = E for both dipole sets at once.  Vectors (3n each) in `w`:
//   0 rD  1 rP  2 zD  3 zP  4 pD  5 pP  6 tD  7 tP (T p)        sums (device double[16] behind them): see below
// stage 0: start from the dipoles in indD / indP (alpha E, or a guess extrapolated from earlier solutions: k_mp_predict):
//          r = E - mu / alpha + T mu  (T mu given in t; zero for mu = alpha E up to rounding of alpha / alpha), z = alpha r, p = z;
//          sums[0,1] = r.z (d, p), sums[4,5] = z.z
// stage 1: Ap = p / alpha - t;  sums[2,3] = p.Ap
// stage 2: mu += a p, r -= a Ap, z = alpha r;  sums[0,1] = r.z (new), sums[4,5] = z.z     (a = sums_old[0,1] / sums[2,3], given)
// stage 3: p = z + b p                                                                      (b given)
// The step lengths are formed on the device from the sums the previous stage left (sums[0,1] r.z of the current residual, [2,3] p.Ap,
// [6,7] r.z of the new residual, [4,5] z.z for the convergence measure).  Stage 4 (one thread) moves the new r.z into place, clears the
// accumulators for the next iteration and forms the convergence measure itself -- debye x the RMS of alpha r, the Reference's epsilon
// (convergeInduceDipolesByDIIS :939-1005) --: sums[12] = epsilon, sums[11] = iterations done, sums[10] = 1 once epsilon is below the target.
// Every kernel of an iteration returns at once when sums[10] is set, so the host may enqueue iterations ahead of looking at the measure
// (solve_mutual): no host round trip per iteration.
__global__ void k_mp_cg(MpArgs a, double* w, int stage, double target, double unused) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n3 = 3 * (size_t) a.n;
    double* sums = w + 8 * n3;
    const double debye = 48.033324;          // AmoebaReferenceMultipoleForce::_debye
    if (stage == 4 || stage == 5) {
        // 5: after stage 0 (the measure of the first guess); 4: end of an iteration
        if (stage == 5 && a.clearGrids) {
            // (launched over all blocks then: the two grids zeroed for the spreading of the first iteration, as stage 3 / 7 do for the later ones)
            const size_t quads = (size_t) a.nx * a.ny * a.nz / 4, total = (size_t) gridDim.x * blockDim.x;
            float4* g0 = (float4*) a.grid; float4* g1 = (float4*) a.grid2;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < quads; k += total) { g0[k] = z; g1[k] = z; }
        }
        if (i == 0 && stage == 5) {
            // what the list builder found travels to the host with the sums (the deferred list check of solve_mutual)
            sums[13] = a.listOverflow != nullptr ? (double) *a.listOverflow : 0.0;
            sums[15] = a.listBuilds != nullptr ? (double) *a.listBuilds : 0.0;
        }
        if (i == 0 && sums[10] == 0.0) {
            if (stage == 4) { sums[0] = sums[6]; sums[1] = sums[7]; sums[11] += 1.0; }
            const double eps = debye * sqrt(fmax(sums[4], sums[5]) / a.n);
            sums[12] = eps;
            if (eps < target && sums[13] == 0.0) sums[10] = 1.0;          // (sums[13]: the list builder's overflow word -- such a solve never 'converges')
            sums[2] = sums[3] = sums[4] = sums[5] = sums[6] = sums[7] = 0.0;
        }
        return;
    }
    if (stage == 6) {
        if (unused != 0.0 && sums[10] == 0.0) return;          // enqueued before the host has seen the convergence word: only once converged
        // After convergence: mu += alpha r, the update the Reference's iteration has already applied when ITS measure (the size of that very
        // update) falls below the target (convergeInduceDipolesByDIIS :939-1005) -- conjugate gradients stop with the residual of the
        // dipoles they return, one such update short.  No field evaluation: r is the residual of the converged dipoles.
        if (i < a.n) {
            const double pol = a.polarity[i];
            store3(a.indD, i, load3(a.indD, i) + pol * load3(w, i));
            store3(a.indP, i, load3(a.indP, i) + pol * load3(w + n3, i));
        }
        return;
    }
    if (stage != 0 && sums[10] != 0.0) return;
    double cD = 0.0, cP = 0.0;
    const bool closing = stage == 7;          // stage 2 that also ends the iteration (the direction update rides in the next spreading launch)
    if (closing) stage = 2;
    // overlap mode (solve_mutual): 8 = stage 3 with b from sums[8, 9] and without the end of the iteration (stage 7 has done that);
    // 9 = stage 1 on t = the RAW real-space field of p (k_mp_dipole_field, cgStage 2), completed here with the reciprocal field -grad phi
    // and the self field; 10 = stage 0 likewise for the first guess
    const bool plainP = stage == 8, fromRaw = stage == 9 || stage == 10;
    if (plainP) stage = 3;
    if (stage == 9) stage = 1;
    if (stage == 10) stage = 0;
    const double selfTerm = (4.0 / 3.0) * a.alpha * a.alpha * a.alpha / MP_SQRT_PI, invK = 1.0 / OMM_ONE_4PI_EPS0_D;
    if (stage == 2) { cD = sums[2] != 0.0 ? sums[0] / sums[2] : 0.0; cP = sums[3] != 0.0 ? sums[1] / sums[3] : 0.0; }
    if (stage == 3) { cD = sums[0] != 0.0 ? sums[6] / sums[0] : 0.0; cP = sums[1] != 0.0 ? sums[7] / sums[1] : 0.0; }
    if (plainP) { cD = sums[8]; cP = sums[9]; }
    __shared__ double red[4][MP_CG_BLOCK / 64];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (i < a.n) {
        const double pol = a.polarity[i], invPol = pol > 0 ? 1.0 / pol : 0.0;
        double* rD = w; double* rP = w + n3; double* zD = w + 2 * n3; double* zP = w + 3 * n3; double* pD = w + 4 * n3; double* pP = w + 5 * n3; double* tD = w + 6 * n3; double* tP = w + 7 * n3;
        if (fromRaw) {
            // t holds the real-space field of the vectors (mu_0, or p): + reciprocal field + self field = T v, as k_mp_dipole_field's own last lines
            const double* fd = a.phiInd + 20 * (size_t) i;
            const double* fp = a.phiIndP + 20 * (size_t) i;
            const V3 vd = stage == 0 ? load3(a.indD, i) : load3(pD, i), vp = stage == 0 ? load3(a.indP, i) : load3(pP, i);
            store3(tD, i, load3(tD, i) - invK * v3(fd[1], fd[2], fd[3]) + selfTerm * vd);
            store3(tP, i, load3(tP, i) - invK * v3(fp[1], fp[2], fp[3]) + selfTerm * vp);
        }
        if (stage == 0) {
            V3 rd = v3(0, 0, 0), rp = v3(0, 0, 0);
            if (pol > 0) {
                rd = load3(a.fieldD, i) - invPol * load3(a.indD, i) + load3(tD, i);
                rp = load3(a.fieldP, i) - invPol * load3(a.indP, i) + load3(tP, i);
            }
            store3(rD, i, rd); store3(rP, i, rp);
            if (!a.precond) { store3(zD, i, pol * rd); store3(zP, i, pol * rp); store3(pD, i, pol * rd); store3(pP, i, pol * rp); s0 = pol * dot(rd, rd); s1 = pol * dot(rp, rp); }
            s2 = pol * pol * dot(rd, rd); s3 = pol * pol * dot(rp, rp);
        }
        else if (stage == 1) {
            const V3 pd = load3(pD, i), pp = load3(pP, i);
            const V3 ad = invPol * pd - (pol > 0 ? load3(tD, i) : v3(0, 0, 0)), ap = invPol * pp - (pol > 0 ? load3(tP, i) : v3(0, 0, 0));
            store3(tD, i, ad); store3(tP, i, ap);            // t now holds A p
            s0 = dot(pd, ad); s1 = dot(pp, ap);
        }
        else if (stage == 2) {
            const V3 rd = load3(rD, i) - cD * load3(tD, i), rp = load3(rP, i) - cP * load3(tP, i);
            store3(a.indD, i, load3(a.indD, i) + cD * load3(pD, i)); store3(a.indP, i, load3(a.indP, i) + cP * load3(pP, i));
            store3(rD, i, rd); store3(rP, i, rp);
            if (!a.precond) { store3(zD, i, pol * rd); store3(zP, i, pol * rp); s0 = pol * dot(rd, rd); s1 = pol * dot(rp, rp); }
            s2 = pol * pol * dot(rd, rd); s3 = pol * pol * dot(rp, rp);
        }
        else {
            const V3 nd = load3(zD, i) + cD * load3(pD, i), np = load3(zP, i) + cP * load3(pP, i);
            store3(pD, i, nd); store3(pP, i, np);
            if (a.gather != nullptr) {
                float* v = a.gather + 6 * (size_t) (a.slotOfAtom != nullptr && a.order != nullptr ? a.slotOfAtom[i] : i);
                v[0] = (float) nd.x; v[1] = (float) nd.y; v[2] = (float) nd.z; v[3] = (float) np.x; v[4] = (float) np.y; v[5] = (float) np.z;
            }
        }
    }
    if ((stage == 3 || closing) && a.clearGrids) {
        // the two grids are dead between the read-back of this iteration's potentials and the spreading of the next one: zeroed here, not
        // by a launch of their own in front of the spreading
        const size_t quads = (size_t) a.nx * a.ny * a.nz / 4, total = (size_t) gridDim.x * blockDim.x;
        float4* g0 = (float4*) a.grid; float4* g1 = (float4*) a.grid2;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        for (size_t k = (size_t) blockIdx.x * blockDim.x + threadIdx.x; k < quads; k += total) { g0[k] = z; g1[k] = z; }
    }
    if (plainP) return;
    if (stage == 3) {
        // the block that finishes last ends the iteration (what stage 4 does as a launch of its own): every block has read the step
        // lengths from the sums before it takes its ticket
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            const unsigned ticket = atomicAdd((unsigned*) &sums[14], 1u);
            if (ticket == gridDim.x - 1) {
                *(unsigned*) &sums[14] = 0u;
                sums[0] = sums[6]; sums[1] = sums[7]; sums[11] += 1.0;
                const double eps = debye * sqrt(fmax(sums[4], sums[5]) / a.n);
                sums[12] = eps;
                sums[2] = sums[3] = sums[4] = sums[5] = sums[6] = sums[7] = 0.0;
                __threadfence();
                if (eps < target && sums[13] == 0.0) sums[10] = 1.0;          // (sums[13]: the list builder's overflow word -- such a solve never 'converges')
            }
        }
        return;
    }
    // one atomic per block and sum: every same-address atomic of a launch is serialised at the memory side
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2); s3 = wave_sum(s3);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s0; red[1][threadIdx.x >> 6] = s1; red[2][threadIdx.x >> 6] = s2; red[3][threadIdx.x >> 6] = s3; }
    __syncthreads();
    if (threadIdx.x == 0) {
        s0 = s1 = s2 = s3 = 0.0;
        for (int k = 0; k < (int) (blockDim.x >> 6); k++) { s0 += red[0][k]; s1 += red[1][k]; s2 += red[2][k]; s3 += red[3][k]; }
        if (stage == 1) { atomicAdd(&sums[2], s0); atomicAdd(&sums[3], s1); }
        else if (stage == 2) { atomicAdd(&sums[6], s0); atomicAdd(&sums[7], s1); atomicAdd(&sums[4], s2); atomicAdd(&sums[5], s3); }
        else { atomicAdd(&sums[0], s0); atomicAdd(&sums[1], s1); atomicAdd(&sums[4], s2); atomicAdd(&sums[5], s3); }
        if (closing) {
            // The block that finishes last ends the iteration: b = r'.z' / r.z for the direction update (k_mp_spread_bricks reads it), the
            // sums rolled over, the convergence measure and word.  The four accumulators are being added to by THIS launch -- at the memory
            // side, while this XCD's L2 may hold a line of them from the loads above: they are read, and cleared, by exchanges.
            __threadfence();
            const unsigned ticket = atomicAdd((unsigned*) &sums[14], 1u);
            if (ticket == gridDim.x - 1) {
                *(unsigned*) &sums[14] = 0u;
                const double rzD = __longlong_as_double((long long) atomicExch((unsigned long long*) &sums[6], 0ull)), rzP = __longlong_as_double((long long) atomicExch((unsigned long long*) &sums[7], 0ull));
                const double zzD = __longlong_as_double((long long) atomicExch((unsigned long long*) &sums[4], 0ull)), zzP = __longlong_as_double((long long) atomicExch((unsigned long long*) &sums[5], 0ull));
                sums[8] = sums[0] != 0.0 ? rzD / sums[0] : 0.0; sums[9] = sums[1] != 0.0 ? rzP / sums[1] : 0.0;
                sums[0] = rzD; sums[1] = rzP; sums[11] += 1.0;
                sums[2] = sums[3] = 0.0;
                const double eps = debye * sqrt(fmax(zzD, zzP) / a.n);
                sums[12] = eps;
                __threadfence();
                if (eps < target && sums[13] == 0.0) sums[10] = 1.0;          // (sums[13]: the list builder's overflow word -- such a solve never 'converges')
            }
        }
    }
}

// (vD, vP) of every atom as six floats at its scan position: what k_mp_dipole_field gathers (see there)
__global__ void k_mp_pack(MpArgs a, const double* __restrict__ vD, const double* __restrict__ vP) {
    if (a.doneFlag != nullptr && (a.doneFlag[0] != 0.0 || a.doneFlag[3] != 0.0)) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    float* v = a.gather + 6 * (size_t) (a.slotOfAtom != nullptr && a.order != nullptr ? a.slotOfAtom[i] : i);
    v[0] = (float) vD[3 * i]; v[1] = (float) vD[3 * i + 1]; v[2] = (float) vD[3 * i + 2]; v[3] = (float) vP[3 * i]; v[4] = (float) vP[3 * i + 1]; v[5] = (float) vP[3 * i + 2];
}

// First guess of the solver from the solutions of earlier calls: mu(t) ~ sum_k c_k mu(t - k dt), k = 1..count, with the caller's
// coefficients.  history: ring of `slots` records of 6n doubles (mu_d, mu_p), `newest` = the slot of mu(t - dt).
// store = 1: the converged dipoles go into slot `newest` instead (called after the solve).
struct HistoryCoeff { double c[OMMHIP_AMOEBA_MAX_HISTORY]; };
__global__ void k_mp_history(MpArgs a, double* history, int slots, int newest, int count, int store, HistoryCoeff coeff) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (a.needDone != nullptr && *a.needDone == 0.0) return;
    const size_t n3 = 3 * (size_t) a.n;
    if (store) {
        double* rec = history + (size_t) newest * 2 * n3;
        store3(rec, i, load3(a.indD, i)); store3(rec + n3, i, load3(a.indP, i));
        return;
    }
    if (!(a.polarity[i] > 0)) return;
    V3 d = v3(0, 0, 0), p = v3(0, 0, 0);
#pragma unroll
    for (int k = 0; k < OMMHIP_AMOEBA_MAX_HISTORY; k++) {
        if (k >= count) break;
        const double c = coeff.c[k];
        const double* rec = history + (size_t) ((newest - k + slots) % slots) * 2 * n3;
        d = d + c * load3(rec, i); p = p + c * load3(rec + n3, i);
    }
    store3(a.indD, i, d); store3(a.indP, i, p);
}

// ------------------------------------------------------------------------------------------------
// Torques on the multipoles -> forces on the atoms that define their frames.  The rule is Tinker's chain rule as restated in
// AmoebaReferenceMultipoleForce::mapTorqueToForceForParticle (:1476-1691): U = z atom, V = x atom, W = y atom (or U x V).
// ------------------------------------------------------------------------------------------------
__global__ void k_mp_torque_to_force(MpArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int4 ax = a.axis[i];
    if (ax.x == 5 || ax.y < 0) return;
    const V3 tq = a.torque2 != nullptr ? load3(a.torque, i) + load3(a.torque2, i) : load3(a.torque, i), xi = position(a, i);
    V3 u = position(a, ax.y) - xi, v, w;
    const double nu = normalize(u);
    if (ax.z >= 0) v = position(a, ax.z) - xi;
    else v = fabs(u.x) < 0.866 ? v3(1, 0, 0) : v3(0, 1, 0);          // a z-only frame without an x atom: the direction the frame itself was completed with
    const double nv = normalize(v);
    if (ax.w >= 0 && (ax.x == 2 || ax.x == 3)) w = position(a, ax.w) - xi; else w = cross(u, v);
    const double nw = normalize(w);
    V3 uv = cross(v, u), uw = cross(w, u), vw = cross(w, v);
    normalize(uv); normalize(uw); normalize(vw);
    const double cuv = dot(u, v), suv = sqrt(1.0 - cuv * cuv), cuw = dot(u, w), suw = sqrt(1.0 - cuw * cuw), cvw = dot(v, w), svw = sqrt(1.0 - cvw * cvw);
    const double du = -dot(u, tq), dv = -dot(v, tq), dw = -dot(w, tq);
    V3 fu = v3(0, 0, 0), fv = v3(0, 0, 0), fw = v3(0, 0, 0);
    if (ax.x == 0 || ax.x == 1) {
        const double half = ax.x == 1 ? 0.5 : 1.0;
        fu = (dv / (nu * suv)) * uv + (half * dw / nu) * uw;
        fv = (-du / (nv * suv)) * uv + (ax.x == 1 ? 0.5 * dw / nv : 0.0) * vw;
    }
    else if (ax.x == 2) {
        V3 rr = v + w, s = cross(u, rr);
        normalize(rr); normalize(s);
        V3 ur = cross(rr, u), us = cross(s, u);
        normalize(ur); normalize(us);
        const double cur = dot(u, rr), sur = sqrt(1.0 - cur * cur), cvs = dot(v, s), svs = sqrt(1.0 - cvs * cvs), cws = dot(w, s), sws = sqrt(1.0 - cws * cws);
        V3 t1 = v - cvs * s, t2 = w - cws * s;
        normalize(t1); normalize(t2);
        const double ut1cos = dot(u, t1), ut1sin = sqrt(1.0 - ut1cos * ut1cos), ut2cos = dot(u, t2), ut2sin = sqrt(1.0 - ut2cos * ut2cos);
        const double dr = -dot(rr, tq), ds = -dot(s, tq);
        fu = (dr / (nu * sur)) * ur + (ds / nu) * us;
        fv = (du / (nv * (ut1sin + ut2sin))) * (svs * s - cvs * t1);
        fw = (du / (nw * (ut1sin + ut2sin))) * (sws * s - cws * t2);
    }
    else if (ax.x == 3) {
        fu = (1.0 / 3.0) * ((dw / (nu * suw)) * uw + (dv / (nu * suv)) * uv - (du / (nu * suw)) * uw - (du / (nu * suv)) * uv);
        fv = (1.0 / 3.0) * ((dw / (nv * svw)) * vw - (du / (nv * suv)) * uv - (dv / (nv * svw)) * vw + (dv / (nv * suv)) * uv);
        fw = (1.0 / 3.0) * ((-du / (nw * suw)) * uw - (dv / (nw * svw)) * vw + (dw / (nw * suw)) * uw + (dw / (nw * svw)) * vw);
    }
    else if (ax.x == 4) fu = (dv / (nu * suv)) * uv + (dw / nu) * uw;
    add_force(a.force, a.paddedAtoms, a.slotOfAtom[ax.y], -fu.x, -fu.y, -fu.z);
    if (ax.x != 4 && ax.z >= 0) add_force(a.force, a.paddedAtoms, a.slotOfAtom[ax.z], -fv.x, -fv.y, -fv.z);
    if ((ax.x == 2 || ax.x == 3) && ax.w >= 0) add_force(a.force, a.paddedAtoms, a.slotOfAtom[ax.w], -fw.x, -fw.y, -fw.z);
    else fw = v3(0, 0, 0);
    if (ax.x == 4) fv = v3(0, 0, 0);
    const V3 fi = fu + fv + fw;
    add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], fi.x, fi.y, fi.z);
}

bool make_args(const ommhip_amoeba_multipole* mp, const void* pos_d, const double box[6], MpArgs& a) {
    const ommhip_pme* pme = (const ommhip_pme*) mp->pme;
    if (mp->num_atoms <= 0 || pme == nullptr || pme->grid_real == nullptr || mp->axis == nullptr || mp->special_start == nullptr) return false;
    a.n = mp->num_atoms; a.nx = pme->nx; a.ny = pme->ny; a.nz = pme->nz;
    a.paddedAtoms = 0; a.includeEnergy = 0; a.energySlots = 1;
    a.pos = (const double4*) pos_d;
    a.charge = mp->charge; a.molDipole = mp->mol_dipole; a.molQuad = mp->mol_quadrupole; a.axis = (const int4*) mp->axis;
    a.thole = mp->thole; a.damping = mp->damping; a.polarity = mp->polarity;
    a.specStart = mp->special_start; a.specAtom = mp->special_atom; a.specScale = (const double4*) mp->special_scale;
    a.cutoff2 = mp->cutoff * mp->cutoff; a.alpha = mp->alpha;
    a.labDipole = mp->lab_dipole; a.labQuad = mp->lab_quadrupole; a.fieldD = mp->field_d; a.fieldP = mp->field_p;
    a.indD = mp->induced_d; a.indP = mp->induced_p; a.phi = mp->phi; a.phiInd = mp->phi_induced; a.torque = mp->torque;
    a.mutual = mp->mutual != 0 ? 1 : 0; a.phiIndP = mp->phi_induced_p;
    if (a.mutual && (mp->phi_induced_p == nullptr || mp->solver == nullptr)) return false;
    a.box.ax = box[0]; a.box.bx = box[1]; a.box.by = box[2]; a.box.cx = box[3]; a.box.cy = box[4]; a.box.cz = box[5];
    // reciprocal box (ReferencePME.cpp:196-204): s_k = sum_c x_c R[c][k]
    const double det = box[0] * box[2] * box[5];
    double R[3][3] = {{box[2] * box[5] / det, 0, 0}, {-box[1] * box[5] / det, box[0] * box[5] / det, 0}, {(box[1] * box[4] - box[2] * box[3]) / det, -box[0] * box[4] / det, box[0] * box[2] / det}};
    const int n[3] = {a.nx, a.ny, a.nz};
    for (int k = 0; k < 3; k++)
        for (int c = 0; c < 3; c++) a.a[k][c] = n[k] * R[c][k];
    a.grid = (float*) pme->grid_real; a.grid2 = nullptr; a.clearGrids = 0;
    a.slotOfAtom = nullptr; a.force = nullptr; a.energyBuffer = nullptr;
    // pair lists: scan positions = the platform's slots when the caller provides the order, atoms otherwise
    if (mp->pair_list == nullptr || mp->pair_count == nullptr || mp->pair_overflow == nullptr || mp->special_pos == nullptr || mp->special_scale_sorted == nullptr || mp->tile_bounds == nullptr || mp->pair_cap < 1) return false;
    a.order = nullptr; a.numScan = a.n;
    if (mp->atom_of_slot != nullptr && mp->slot_of_atom != nullptr && mp->scan_slots >= a.n) { a.order = mp->atom_of_slot; a.slotOfAtom = mp->slot_of_atom; a.numScan = mp->scan_slots; }
    a.pairList = mp->pair_list; a.pairCount = mp->pair_count; a.listStride = a.numScan; a.listSubcap = mp->pair_cap / PL_PARTS;
    if (a.listSubcap < 1) return false;
    a.specScaleSorted = (const double4*) mp->special_scale_sorted;
    a.pairCache = a.mutual ? mp->pair_cache : nullptr; a.pairCap = a.listSubcap * PL_PARTS;
    // opt-in (OPENMM_HIP_AMOEBA_PRECOND=1): on the 36 501-atom water box it saves one iteration of eight (epsilon 1e-5) and costs a list walk
    // per iteration -- no gain; the default stays z = alpha r
    static const bool usePrecond = getenv("OPENMM_HIP_AMOEBA_PRECOND") != nullptr && atoi(getenv("OPENMM_HIP_AMOEBA_PRECOND")) != 0;
    a.precond = a.pairCache != nullptr && usePrecond ? 1 : 0; a.precondCut2 = 0.45 * 0.45;
    a.specialAdds = 0; a.pairsOnly = 0; a.torque2 = nullptr; a.needDone = nullptr;
    a.listOverflow = mp->pair_overflow; a.listBuilds = mp->skin > 0.0 && mp->ref_pos != nullptr && mp->list_state != nullptr ? mp->list_state + 2 : nullptr;
    a.doneFlag = nullptr;
    a.gather = a.mutual ? mp->solver_gather : nullptr;
    return true;
}

int spread_blocks(const MpArgs& a) { return (int) (((size_t) a.n * MP_SPREAD_LANES + 255) / 256); }

// number of workgroups of the pair kernels (MP_SPLIT threads per scan position)
int scan_blocks(const MpArgs& a) { return (int) (((size_t) a.numScan * MP_SPLIT + MP_BLOCK - 1) / MP_BLOCK); }

// the pair lists of this evaluation (amoeba_pairs.h); -2: the lists did not fit into pair_cap entries per atom (*pair_needed says how many would)
// two pinned host words per call for the deferred list check (overflow word, build counter); a small ring: one call is in flight per host thread
// (portable pinned memory: the ring is shared by the Contexts of every device of the process)
#define OMMHIP_MP_MAX_DEVICES 64
int* deferred_words() {
    static int* const ring = [] { int* r = nullptr; return hipHostMalloc((void**) &r, sizeof(int) * 2 * 16, hipHostMallocPortable) == hipSuccess ? r : (int*) nullptr; }();
    static std::atomic<unsigned> next(0);
    if (ring == nullptr) return nullptr;
    int* w = ring + 2 * (next.fetch_add(1) % 16);
    w[0] = 0; w[1] = 0;
    return w;
}

// sixteen pinned doubles per call for the solver's sums when they are read behind an event (a small ring, as above)
double* pinned_sums() {
    static double* const ring = [] { double* r = nullptr; return hipHostMalloc((void**) &r, sizeof(double) * 16 * 8, hipHostMallocPortable) == hipSuccess ? r : (double*) nullptr; }();
    static std::atomic<unsigned> next(0);
    return ring == nullptr ? nullptr : ring + 16 * (next.fetch_add(1) % 8);
}

template <class Between>
int build_pair_lists(const ommhip_amoeba_multipole* mp, const MpArgs& a, const double box[6], hipStream_t st, Between between, int* deferred = nullptr, bool deferredCopies = true) {
    PairListArgs p;
    p.n = a.n; p.numScan = a.numScan; p.subcap = a.listSubcap; p.stride = a.numScan; p.excludeListed = 0;
    static const bool noTiles = getenv("OPENMM_HIP_AMOEBA_NO_TILES") != nullptr;         // A/B knob: the builder looks at every tile
    p.skipTiles = !noTiles && box[1] == 0.0 && box[3] == 0.0 && box[4] == 0.0 ? 1 : 0;
    p.pos = a.pos; p.order = a.order; p.slotOfAtom = a.slotOfAtom; p.box = a.box; p.cutoff2 = a.cutoff2;
    const int tiles = (a.numScan + PL_BLOCK - 1) / PL_BLOCK;
    p.tileCenter = (double4*) mp->tile_bounds; p.tileHalf = p.tileCenter + tiles;
    p.rowStart = a.specStart; p.rowAtom = a.specAtom; p.rowPos = mp->special_pos;
    p.rowData = (double4*) mp->special_scale_sorted; p.rowDataIn = a.specScale;
    p.list = mp->pair_list; p.count = mp->pair_count; p.overflow = mp->pair_overflow;
    // Verlet skin: the list holds the partners within cutoff + skin and lives until an atom has moved by skin / 2 (the pair kernels re-test)
    p.refPos = nullptr; p.state = nullptr; p.skinHalf2 = 0.0; p.forceRebuild = 1;
    if (mp->skin > 0.0 && mp->ref_pos != nullptr && mp->list_state != nullptr) {
        const double radius = mp->cutoff + mp->skin;
        p.cutoff2 = radius * radius; p.refPos = (double4*) mp->ref_pos; p.state = mp->list_state; p.skinHalf2 = 0.25 * mp->skin * mp->skin; p.forceRebuild = mp->force_rebuild != 0;
    }
    return pl_launch(p, mp->pair_needed, st, mp->list_builds, between, deferred, deferredCopies);
}
int build_pair_lists(const ommhip_amoeba_multipole* mp, const MpArgs& a, const double box[6], hipStream_t st) { return build_pair_lists(mp, a, box, st, [] {}); }

// frames, reciprocal potential of the permanent multipoles, fields and induced dipoles
int launch_induce(const ommhip_amoeba_multipole* mp, const MpArgs& a, const double box[6], hipStream_t st, bool hook = false, int* deferred = nullptr) {
    const ommhip_pme* pme = (const ommhip_pme*) mp->pme;
    // Frames and the reciprocal potential of the permanent multipoles need no lists.  With the side stream of the mutual solver at hand
    // (stream2, event_a / event_b; round 5) that chain -- clear, spread, three transform launches, read-back: ~110 us of small launches on
    // DHFR -- runs BESIDE the list build (190 us at one or two wavefronts per SIMD) and the covalently-related pairs, and the main stream
    // waits for it in front of the field kernel, whose last lines read the potential.  Without one it is enqueued behind the builder's kernels
    // BEFORE the host waits for the builder's overflow word.  Either way the caller's hook -- the platform launches its AmoebaVdwForce there,
    // whose own list build then runs beside this one -- gets its turn before that wait.  A call that returns -2 has written work arrays only.
    static const bool hookFirst = getenv("OPENMM_HIP_AMOEBA_HOOK_LAST") == nullptr;       // A/B
    static const bool noSide = getenv("OPENMM_HIP_AMOEBA_RECIPROCAL_INLINE") != nullptr;   // A/B: the chain on the main stream
    hipStream_t st2 = (hipStream_t) mp->stream2;
    const bool side = !noSide && st2 != nullptr && mp->event_a != nullptr && mp->event_b != nullptr;
    const int blocks = (a.n + MP_BLOCK - 1) / MP_BLOCK;
    auto reciprocal = [&](hipStream_t s) {
        hipMemsetAsync(a.grid, 0, sizeof(float) * (size_t) a.nx * a.ny * a.nz, s);
        hipLaunchKernelGGL(k_mp_spread<false>, dim3(spread_blocks(a)), dim3(256), 0, s, a, (const double*) nullptr, 0.0, (const double*) nullptr, 0.0);
        ommhip_pme_convolve(pme, s);
        hipLaunchKernelGGL(k_mp_potential<3>, dim3(spread_blocks(a)), dim3(256), 0, s, a, a.phi, (double*) nullptr);
    };
    hipLaunchKernelGGL(k_mp_frames, dim3(blocks), dim3(MP_BLOCK), 0, st, a);
    if (side) {
        hipEventRecord((hipEvent_t) mp->event_a, st);                  // the lab-frame multipoles are there
        hipStreamWaitEvent(st2, (hipEvent_t) mp->event_a, 0);
        reciprocal(st2);
        hipEventRecord((hipEvent_t) mp->event_b, st2);
    }
    // deferred (round 5): the builder's overflow word is looked at after the caller's next wait on this stream -- the solver's first look at its
    // convergence measure -- instead of by a wait of its own here: everything up to the force kernels writes work arrays only.  The hook then
    // runs at the end of this function, with the list build, the field kernels and the side chain already enqueued.
    const int rc = build_pair_lists(mp, a, box, st, [&] {
        if (deferred == nullptr && hook && mp->after_lists_enqueued != nullptr && hookFirst) mp->after_lists_enqueued(mp->after_lists_arg);
        if (!side) reciprocal(st);
        if (deferred == nullptr && hook && mp->after_lists_enqueued != nullptr && !hookFirst) mp->after_lists_enqueued(mp->after_lists_arg);
    }, deferred, !a.mutual);          // (mutual polarization: the solver's stage 5 puts the two words among its sums)
    if (rc != 0) { if (side) hipStreamWaitEvent(st, (hipEvent_t) mp->event_b, 0); return rc; }      // (the side chain does not outlive the call)
    if (mp->mixed_precision) hipLaunchKernelGGL(k_mp_special<false>, dim3((unsigned) (((size_t) a.n * MP_SPLIT + MP_BLOCK - 1) / MP_BLOCK)), dim3(MP_BLOCK), 0, st, a);
    // the field kernel's last lines read the reciprocal potential: with the side chain they are a launch of their own behind the wait, and the
    // pair sums -- 127 us on DHFR -- run beside the chain as well
    MpArgs b = a;
    b.pairsOnly = side ? 1 : 0;
    if (mp->mixed_precision) hipLaunchKernelGGL(k_mp_field<true>, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, b);
    else hipLaunchKernelGGL(k_mp_field<false>, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, b);
    if (side) {
        hipStreamWaitEvent(st, (hipEvent_t) mp->event_b, 0);
        hipLaunchKernelGGL(k_mp_field_finish, dim3(blocks), dim3(MP_BLOCK), 0, st, a);
    }
    if (deferred != nullptr && hook && mp->after_lists_enqueued != nullptr) mp->after_lists_enqueued(mp->after_lists_arg);
    return 0;
}

// the deferred list check, after a wait on the stream the lists were built on: 0, or -2 as build_pair_lists returns it
int deferred_lists_result(const ommhip_amoeba_multipole* mp, const int* deferred) {
    const int rc = pl_deferred_result(deferred, mp->pair_needed, mp->list_builds, mp->skin > 0.0 && mp->ref_pos != nullptr && mp->list_state != nullptr);
    static const bool report = getenv("OPENMM_HIP_AMOEBA_DEBUG") != nullptr;
    if (report && rc == -2) fprintf(stderr, "amoeba lists: did not fit (found at the deferred check): %d entries per atom needed\n", mp->pair_needed != nullptr ? *mp->pair_needed : -1);
    return rc;
}

// potential (and derivatives) of one set of dipoles at the atoms
// spread of one set of induced dipoles: LDS bricks in the slot order, the 8-lanes-per-atom kernel without one (or OPENMM_HIP_AMOEBA_NO_BRICKS, the A/B knob)
void spread_induced(const MpArgs& a, const double* A, double sA, const double* B, double sB, hipStream_t st) {
    static const bool noBricks = getenv("OPENMM_HIP_AMOEBA_NO_BRICKS") != nullptr;
    if (a.order != nullptr && !noBricks && a.nx >= MPB_BRICK && a.ny >= MPB_BRICK && a.nz >= MPB_BRICK)
        hipLaunchKernelGGL(k_mp_spread_bricks, dim3((a.numScan + MPB_ATOMS - 1) / MPB_ATOMS), dim3(256), 0, st, a, A, sA, B, sB, (const double*) nullptr, (double*) nullptr);
    else
        hipLaunchKernelGGL(k_mp_spread<true>, dim3(spread_blocks(a)), dim3(256), 0, st, a, A, sA, B, sB);
}

// maxOrder: derivatives of the potential wanted (1: gradient, a solver iteration; 2: + second derivatives, an order of the extrapolation; 3: all)
void dipole_potential(const ommhip_pme* pme, const MpArgs& a, const double* dipoles, double* out, hipStream_t st, int maxOrder = 3) {
    hipMemsetAsync(a.grid, 0, sizeof(float) * (size_t) a.nx * a.ny * a.nz, st);
    spread_induced(a, dipoles, 1.0, nullptr, 0.0, st);
    ommhip_pme_convolve(pme, st);
    if (maxOrder == 1) hipLaunchKernelGGL(k_mp_potential<1>, dim3(spread_blocks(a)), dim3(256), 0, st, a, out, (double*) nullptr);
    else if (maxOrder == 2) hipLaunchKernelGGL(k_mp_potential<2>, dim3(spread_blocks(a)), dim3(256), 0, st, a, out, (double*) nullptr);
    else hipLaunchKernelGGL(k_mp_potential<3>, dim3(spread_blocks(a)), dim3(256), 0, st, a, out, (double*) nullptr);
}

// The potentials of two sets of dipoles.  The chain of one set -- clear, spread, three transform launches, read-back -- is six small launches
// that leave most of the chip idle; with a second grid and a side stream (pme2, stream2, two ordering events: optional in the C ABI) the
// chain of the second set runs beside that of the first.
// fieldOnly: potential and gradient only (a solver iteration); the derivatives up to third order are computed for the converged dipoles
// Do the two sets of induced dipoles travel through the same launches (dipole_potentials)?
bool two_grid_launches(const ommhip_amoeba_multipole* mp, const MpArgs& a) {
    const ommhip_pme* pme = (const ommhip_pme*) mp->pme;
    const ommhip_pme* pme2 = (const ommhip_pme*) mp->pme2;
    static const bool twoStreams = getenv("OPENMM_HIP_AMOEBA_TWO_STREAMS") != nullptr && getenv("OPENMM_HIP_AMOEBA_TWO_STREAMS")[0] == '1';
    static const bool noBricks = getenv("OPENMM_HIP_AMOEBA_NO_BRICKS") != nullptr;
    if (pme2 == nullptr || pme2->grid_real == nullptr || pme2->grid_real == pme->grid_real) return false;
    return !twoStreams && !noBricks && a.order != nullptr && a.nx >= MPB_BRICK && a.ny >= MPB_BRICK && a.nz >= MPB_BRICK && pme->nx == pme2->nx && pme->ny == pme2->ny &&
           pme->nz == pme2->nz && pme->nz * (pme->ny + 1) <= 9472 && pme->fft_mode != 1 && ((size_t) a.nx * a.ny * a.nz) % 4 == 0;
}

// precleared: both grids are zero already (k_mp_cg stage 3 of the iteration before, two-grid launches only: two_grid_launches())
void dipole_potentials(const ommhip_amoeba_multipole* mp, const MpArgs& a, const double* vD, double* outD, const double* vP, double* outP, hipStream_t st, bool fieldOnlyFlag = false, int maxOrder = 0, bool precleared = false, double* cgw = nullptr) {
    const int fieldOnly = maxOrder > 0 ? maxOrder : (fieldOnlyFlag ? 1 : 3);
    const ommhip_pme* pme = (const ommhip_pme*) mp->pme;
    const ommhip_pme* pme2 = (const ommhip_pme*) mp->pme2;
    if (pme2 == nullptr || mp->stream2 == nullptr || mp->event_a == nullptr || mp->event_b == nullptr || pme2->grid_real == nullptr || pme2->grid_real == pme->grid_real) {
        dipole_potential(pme, a, vD, outD, st, fieldOnly);
        dipole_potential(pme, a, vP, outP, st, fieldOnly);
        return;
    }
    // Both chains in the SAME launches (blockIdx.y picks the set of dipoles and its grid): six launches per pair of potentials instead of
    // twelve on two streams with two ordering events.  OPENMM_HIP_AMOEBA_TWO_STREAMS=1 restores the two streams (A/B).
    if (two_grid_launches(mp, a)) {
        MpArgs b = a;
        b.grid2 = (float*) pme2->grid_real;
        const size_t gridBytes = sizeof(float) * (size_t) a.nx * a.ny * a.nz;
        if (precleared) { }
        else if ((gridBytes & 15) == 0) ommhip_clear2(b.grid, gridBytes, b.grid2, gridBytes, (void*) st);          // both grids in one launch
        else { hipMemsetAsync(b.grid, 0, gridBytes, st); hipMemsetAsync(b.grid2, 0, gridBytes, st); }
        hipLaunchKernelGGL(k_mp_spread_bricks, dim3((a.numScan + MPB_ATOMS - 1) / MPB_ATOMS, 2), dim3(256), 0, st, b, vD, 1.0, (const double*) nullptr, 0.0, vP, cgw);
        if (ommhip_pme_convolve2(pme, pme2, st) == 0) {
            if (fieldOnly == 1) hipLaunchKernelGGL(k_mp_potential<1>, dim3(spread_blocks(a), 2), dim3(256), 0, st, b, outD, outP);
            else if (fieldOnly == 2) hipLaunchKernelGGL(k_mp_potential<2>, dim3(spread_blocks(a), 2), dim3(256), 0, st, b, outD, outP);
            else hipLaunchKernelGGL(k_mp_potential<3>, dim3(spread_blocks(a), 2), dim3(256), 0, st, b, outD, outP);
            return;
        }
        // (shape not covered by the two-grid transform: the grids are spread already -- finish them one after the other)
        ommhip_pme_convolve(pme, st);
        ommhip_pme_convolve(pme2, st);
        if (fieldOnly == 1) hipLaunchKernelGGL(k_mp_potential<1>, dim3(spread_blocks(a), 2), dim3(256), 0, st, b, outD, outP);
        else if (fieldOnly == 2) hipLaunchKernelGGL(k_mp_potential<2>, dim3(spread_blocks(a), 2), dim3(256), 0, st, b, outD, outP);
        else hipLaunchKernelGGL(k_mp_potential<3>, dim3(spread_blocks(a), 2), dim3(256), 0, st, b, outD, outP);
        return;
    }
    hipStream_t st2 = (hipStream_t) mp->stream2;
    MpArgs a2 = a;
    a2.grid = (float*) pme2->grid_real;
    hipEventRecord((hipEvent_t) mp->event_a, st);                  // the dipoles (and everything else the side chain reads) are ready
    hipStreamWaitEvent(st2, (hipEvent_t) mp->event_a, 0);
    dipole_potential(pme2, a2, vP, outP, st2, fieldOnly);
    hipEventRecord((hipEvent_t) mp->event_b, st2);
    dipole_potential(pme, a, vD, outD, st, fieldOnly);
    hipStreamWaitEvent(st, (hipEvent_t) mp->event_b, 0);           // the main stream goes on when both potentials are there
}

// Mutual polarization: conjugate gradients from the direct-polarization dipoles, or from a guess extrapolated from the solutions of the
// previous calls (mp->history_use > 0).  Leaves mu_d, mu_p and their potentials (phiInd, phiIndP).
// The convergence measure is formed on the device (k_mp_cg stage 4 / 5) and every kernel of an iteration gives up at once when it has
// been met, so the host enqueues mp->expected_iterations - 1 iterations (what the previous call needed; 0 = unknown) before it first waits
// for the measure, then one at a time: two host round trips per solve instead of one per iteration.  (The FFT launches of an iteration
// enqueued in vain -- the call needed fewer iterations than the one before -- still run, on cleared grids.)
// mainBesideFinal (finalOnSide only): called when everything in front of the final potentials has been enqueued on `st` and before the side chain
// is -- the caller launches THERE what the main stream runs beside that chain (the pair force kernel), so that it does not queue on the host
// behind the side chain's seven launches (50 us in the r11ax timeline).
template <class MainBesideFinal>
int solve_mutual(const ommhip_amoeba_multipole* mp, MpArgs a, hipStream_t st, bool finalOnSide, int* deferred, MainBesideFinal mainBesideFinal) {
    const int blocks = (a.n + MP_BLOCK - 1) / MP_BLOCK;
    const size_t n3 = 3 * (size_t) a.n;
    double* w = mp->solver;
    double* sums = w + 8 * n3;
    double* tD = w + 6 * n3; double* tP = w + 7 * n3; double* pD = w + 4 * n3; double* pP = w + 5 * n3;
    double h[16] = {0};
    bool listsChecked = deferred == nullptr;
    hipStream_t sumsStream = st;              // the stream whose kernels form the sums (the side stream while it carries the iterations' spine)
    auto readSums = [&]() -> int {
        hipError_t e = hipMemcpyAsync(h, sums, sizeof(double) * 16, hipMemcpyDeviceToHost, sumsStream);
        if (e != hipSuccess) return (int) e;
        e = hipStreamSynchronize(sumsStream);
        if (e != hipSuccess) return (int) e;
        // the first wait of the call on this stream: what the list builder found (deferred check) -- lists that did not fit end the call here,
        // before anything has been added to the forces or the history
        if (!listsChecked) { listsChecked = true; deferred[0] = (int) h[13]; deferred[1] = (int) h[15]; return deferred_lists_result(mp, deferred); }
        return 0;
    };
    const bool haveHistory = mp->history != nullptr && mp->history_slots >= 1;
    const int use = haveHistory ? (mp->history_use < 0 ? 0 : (mp->history_use > OMMHIP_AMOEBA_MAX_HISTORY ? OMMHIP_AMOEBA_MAX_HISTORY : (mp->history_use > mp->history_slots ? mp->history_slots : mp->history_use))) : 0;
    HistoryCoeff coeff;
    for (int k = 0; k < OMMHIP_AMOEBA_MAX_HISTORY; k++) coeff.c[k] = mp->history_coeff[k];
    hipMemsetAsync(sums, 0, sizeof(double) * 16, st);
    if (use > 0) {
        const int newest = (mp->history_newest % mp->history_slots + mp->history_slots) % mp->history_slots;
        hipLaunchKernelGGL(k_mp_history, dim3(blocks), dim3(MP_BLOCK), 0, st, a, mp->history, mp->history_slots, newest, use, 0, coeff);
    }
    // T mu_0 and the residual of the first guess
    const int cgBlocks = (a.n + MP_CG_BLOCK - 1) / MP_CG_BLOCK;
    static const bool noOverlap0 = getenv("OPENMM_HIP_AMOEBA_NO_OVERLAP") != nullptr;
    const bool overlap0 = !noOverlap0 && !a.precond && a.gather != nullptr && mp->stream2 != nullptr && mp->event_a != nullptr && mp->event_b != nullptr && two_grid_launches(mp, a);
    if (overlap0) {
        // as an iteration in overlap mode (below): the reciprocal chain of mu_0 on the side stream, its real-space pairs on this one, stage 10 = stage 0
        // on the two parts put together
        hipStream_t s2 = (hipStream_t) mp->stream2;
        hipEventRecord((hipEvent_t) mp->event_a, st);                  // mu_0 is there (the direct dipoles, or the predictor's)
        hipStreamWaitEvent(s2, (hipEvent_t) mp->event_a, 0);
        dipole_potentials(mp, a, a.indD, a.phiInd, a.indP, a.phiIndP, s2, true);
        hipEventRecord((hipEvent_t) mp->event_b, s2);
        hipLaunchKernelGGL(k_mp_pack, dim3(blocks), dim3(MP_BLOCK), 0, st, a, a.indD, a.indP);
        hipLaunchKernelGGL(k_mp_dipole_field, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, a.indD, a.indP, a.phiInd, a.phiIndP, tD, tP, w, 2);
        hipStreamWaitEvent(st, (hipEvent_t) mp->event_b, 0);
        hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, a, w, 10, 0.0, 0.0);
    }
    else {
    dipole_potentials(mp, a, a.indD, a.phiInd, a.indP, a.phiIndP, st, true);
    if (a.gather != nullptr) hipLaunchKernelGGL(k_mp_pack, dim3(blocks), dim3(MP_BLOCK), 0, st, a, a.indD, a.indP);
    }
    if (overlap0) { }
    else if (a.precond) {
        hipLaunchKernelGGL(k_mp_dipole_field, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, a.indD, a.indP, a.phiInd, a.phiIndP, tD, tP, w, -1);
        hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, a, w, 0, 0.0, 0.0);
        hipLaunchKernelGGL(k_mp_precond, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, w, 1);
    }
    else hipLaunchKernelGGL(k_mp_dipole_field, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, a.indD, a.indP, a.phiInd, a.phiIndP, tD, tP, w, 0);
    // two-grid launches: stage 3 of every iteration leaves the grids zeroed for the spreading of the next one
    static const bool clearLaunch = getenv("OPENMM_HIP_AMOEBA_CLEAR_LAUNCH") != nullptr;   // A/B: the clear as its own launch
    const bool clearInStage3 = two_grid_launches(mp, a) && !clearLaunch;
    MpArgs aClear = a;
    if (clearInStage3) { aClear.grid2 = (float*) ((const ommhip_pme*) mp->pme2)->grid_real; aClear.clearGrids = 1; }
    // The direction update p = z + b p of iteration k rides in the spreading launch of iteration k + 1 (two-grid launches, no preconditioner):
    // stage 7 = stage 2 + the grid clear + the end of the iteration; seven launches per iteration instead of eight.  The first iteration's
    // spreading takes the same path with b = 0 (the sums start cleared): it packs p = z for the field kernel's gather, and stage 5 has
    // zeroed the grids for it.
    static const bool noFold = getenv("OPENMM_HIP_AMOEBA_STAGE3_LAUNCH") != nullptr;       // A/B: stage 3 as its own launch
    const bool fold = clearInStage3 && !a.precond && !noFold && a.gather != nullptr;
    if (fold) hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, aClear, w, 5, mp->target_epsilon, 0.0);
    else {
        hipLaunchKernelGGL(k_mp_cg, dim3(1), dim3(64), 0, st, a, w, 5, mp->target_epsilon, 0.0);
        if (a.gather != nullptr) hipLaunchKernelGGL(k_mp_pack, dim3(blocks), dim3(MP_BLOCK), 0, st, a, pD, pP);      // p of the first iteration (later ones: stage 3 writes the copy itself)
    }
    a.doneFlag = sums + 10;                   // from here on the kernels look at the convergence word
    aClear.doneFlag = a.doneFlag;
    static const bool everyIteration = getenv("OPENMM_HIP_AMOEBA_CHECK_EVERY_ITERATION") != nullptr;       // A/B knob: one host round trip per iteration, as before
    static const int enqueueSlack = getenv("OPENMM_HIP_AMOEBA_ENQUEUE_SLACK") != nullptr ? atoi(getenv("OPENMM_HIP_AMOEBA_ENQUEUE_SLACK")) : 0;       // 0: as many as the last solve took (round 5; 1 before: -0.7 %, profiles/r11/r11ag_*)
    const int unchecked = everyIteration ? 0 : (mp->expected_iterations > enqueueSlack ? mp->expected_iterations - enqueueSlack : 0);
    int rc = 0, enqueued = 0;
    bool done = false;
    if (unchecked == 0) { rc = readSums(); if (rc != 0) return rc; done = h[10] != 0.0; }
    // Overlap mode (round 5): the reciprocal-space chain of an iteration -- spread, three transform launches, read-back: ~85 us of small launches --
    // on the side stream, the real-space pairs of k_mp_dipole_field (~90 us at one or two wavefronts per SIMD) on this one, both started by the
    // direction update (stage 8) and met by stage 9, which puts the two parts of T p together and forms p . A p; stage 7 ends the iteration.
    static const bool noOverlap = getenv("OPENMM_HIP_AMOEBA_NO_OVERLAP") != nullptr;       // A/B: the chain of round 5's first half (fold)
    hipStream_t st2 = (hipStream_t) mp->stream2;
    const bool overlap = fold && !noOverlap && st2 != nullptr && mp->event_a != nullptr && mp->event_b != nullptr;
    // Which stream carries the iteration's spine -- direction update, reciprocal chain, vector stages -- and which the pair kernel alone?  The
    // reciprocal chain is the longer half (r11as timeline: 106 against 70 us): with the spine on the SIDE stream the vector stages follow the chain
    // in stream order and the wait for the pair kernel's event finds it signalled already -- the ~14 us of join latency per iteration leave the
    // critical path, and the small launches run at the side stream's priority.  OPENMM_HIP_AMOEBA_SPINE_MAIN=1: the spine on the main stream.
    static const bool spineMain = getenv("OPENMM_HIP_AMOEBA_SPINE_MAIN") != nullptr;
    const bool spineSide = overlap && !spineMain;
    static const bool noSpeculation = getenv("OPENMM_HIP_AMOEBA_NO_SPECULATIVE_TAIL") != nullptr;       // A/B
    static const bool noPolish = getenv("OPENMM_HIP_AMOEBA_NO_POLISH") != nullptr;        // A/B knob
    // One event per host thread AND device: an event belongs to the device that was current when it was created, and a thread may drive
    // Contexts on several GPUs (DeviceIndex); recording it on another device's stream fails.
    static thread_local hipEvent_t sumsEvents[OMMHIP_MP_MAX_DEVICES] = {};
    hipEvent_t sumsEvent = nullptr;
    {
        int dev = -1;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < OMMHIP_MP_MAX_DEVICES) {
            if (sumsEvents[dev] == nullptr && hipEventCreateWithFlags(&sumsEvents[dev], hipEventDisableTiming) != hipSuccess) sumsEvents[dev] = nullptr;
            sumsEvent = sumsEvents[dev];
        }
    }
    const bool speculate = spineSide && finalOnSide && !noSpeculation && sumsEvent != nullptr && unchecked > 0;
    bool tailEnqueued = false;
    if (spineSide && !done && enqueued < mp->max_iterations) {
        hipEventRecord((hipEvent_t) mp->event_a, st);                  // the residual of the first guess, stage 5
        hipStreamWaitEvent(st2, (hipEvent_t) mp->event_a, 0);
        sumsStream = st2;
        while (!done && enqueued < mp->max_iterations) {
            hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st2, a, w, 8, 0.0, 0.0);         // p = z + b p (b = 0 at first), packed for the gather
            hipEventRecord((hipEvent_t) mp->event_a, st2);
            hipStreamWaitEvent(st, (hipEvent_t) mp->event_a, 0);
            hipLaunchKernelGGL(k_mp_dipole_field, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, pD, pP, a.phiInd, a.phiIndP, tD, tP, w, 2);      // real-space field of p, raw
            hipEventRecord((hipEvent_t) mp->event_b, st);
            dipole_potentials(mp, a, pD, a.phiInd, pP, a.phiIndP, st2, true, 0, true);                         // (grids zeroed by stage 5 / the last stage 7)
            hipStreamWaitEvent(st2, (hipEvent_t) mp->event_b, 0);
            hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st2, a, w, 9, 0.0, 0.0);         // T p complete, A p, p . A p
            hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st2, aClear, w, 7, mp->target_epsilon, 0.0);
            enqueued++;
            if (speculate && !tailEnqueued && enqueued >= unchecked && enqueued < mp->max_iterations) {
                // The first look at the sums, with the TAIL of the solve enqueued in front of it (round 5): the sums travel to pinned memory behind
                // an event of their own, and while the host waits for that event the device already has the last update, the history record and
                // the chain of the converged dipoles' potentials in its queues -- all of them kernels that leave at once unless the
                // convergence word is set (stage 6 with its last argument, MpArgs::needDone).  Converged (the usual case: as many iterations
                // as the last solve took): the 40 us the host needs to notice are no longer on the critical path.  Not converged: the tail
                // has done nothing, the loop goes on and the tail is enqueued again, the ordinary way, at the end.
                double* const hp = pinned_sums();
                // (a copy or an event that could not be enqueued: the ordinary, synchronous look at the sums below)
                if (hp != nullptr && hipMemcpyAsync(hp, sums, sizeof(double) * 16, hipMemcpyDeviceToHost, st2) == hipSuccess && hipEventRecord(sumsEvent, st2) == hipSuccess) {
                    hipEventRecord((hipEvent_t) mp->event_a, st2);             // the main stream goes on behind the spine
                    hipStreamWaitEvent(st, (hipEvent_t) mp->event_a, 0);
                    MpArgs t = a;
                    t.doneFlag = nullptr; t.needDone = sums + 10;
                    if (!noPolish) hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, t, w, 6, 0.0, 1.0);
                    if (haveHistory && mp->history_store >= 0)
                        hipLaunchKernelGGL(k_mp_history, dim3(blocks), dim3(MP_BLOCK), 0, st, t, mp->history, mp->history_slots, mp->history_store % mp->history_slots, 0, 1, coeff);
                    hipEventRecord((hipEvent_t) mp->event_a, st);
                    hipStreamWaitEvent(st2, (hipEvent_t) mp->event_a, 0);
                    dipole_potentials(mp, t, a.indD, a.phiInd, a.indP, a.phiIndP, st2, false, 0, true);
                    hipEventRecord((hipEvent_t) mp->event_b, st2);
                    const hipError_t e = hipEventSynchronize(sumsEvent);
                    if (e != hipSuccess) return (int) e;
                    for (int k = 0; k < 16; k++) h[k] = hp[k];
                    if (!listsChecked) { listsChecked = true; deferred[0] = (int) h[13]; deferred[1] = (int) h[15]; rc = deferred_lists_result(mp, deferred); if (rc != 0) return rc; }
                    done = h[10] != 0.0;
                    tailEnqueued = done;
                    continue;
                }
            }
            if (enqueued >= unchecked || enqueued == mp->max_iterations) { rc = readSums(); if (rc != 0) return rc; done = h[10] != 0.0; }
        }
        sumsStream = st;
        if (!tailEnqueued) {
            hipEventRecord((hipEvent_t) mp->event_a, st2);             // the main stream goes on behind the spine
            hipStreamWaitEvent(st, (hipEvent_t) mp->event_a, 0);
        }
    }
    while (!done && enqueued < mp->max_iterations && overlap) {
        hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, a, w, 8, 0.0, 0.0);              // p = z + b p (b = 0 at first), packed for the gather
        hipEventRecord((hipEvent_t) mp->event_a, st);
        hipStreamWaitEvent(st2, (hipEvent_t) mp->event_a, 0);
        dipole_potentials(mp, a, pD, a.phiInd, pP, a.phiIndP, st2, true, 0, true);                             // (grids zeroed by stage 5 / the last stage 7)
        hipEventRecord((hipEvent_t) mp->event_b, st2);
        hipLaunchKernelGGL(k_mp_dipole_field, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, pD, pP, a.phiInd, a.phiIndP, tD, tP, w, 2);      // real-space field of p, raw
        hipStreamWaitEvent(st, (hipEvent_t) mp->event_b, 0);
        hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, a, w, 9, 0.0, 0.0);              // T p complete, A p, p . A p
        hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, aClear, w, 7, mp->target_epsilon, 0.0);
        enqueued++;
        if (enqueued >= unchecked || enqueued == mp->max_iterations) { rc = readSums(); if (rc != 0) return rc; done = h[10] != 0.0; }
    }
    while (!done && enqueued < mp->max_iterations && fold) {
        dipole_potentials(mp, a, pD, a.phiInd, pP, a.phiIndP, st, true, 0, true, w);
        hipLaunchKernelGGL(k_mp_dipole_field, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, pD, pP, a.phiInd, a.phiIndP, tD, tP, w, 1);      // T p, Ap, p.Ap
        hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, aClear, w, 7, mp->target_epsilon, 0.0);
        enqueued++;
        if (enqueued >= unchecked || enqueued == mp->max_iterations) { rc = readSums(); if (rc != 0) return rc; done = h[10] != 0.0; }
    }
    while (!done && enqueued < mp->max_iterations) {
        dipole_potentials(mp, a, pD, a.phiInd, pP, a.phiIndP, st, true, 0, clearInStage3 && enqueued > 0);
        if (a.precond) {
            hipLaunchKernelGGL(k_mp_dipole_field, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, pD, pP, a.phiInd, a.phiIndP, tD, tP, w, -1);
            hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, a, w, 1, 0.0, 0.0);      // Ap, p.Ap
        }
        else hipLaunchKernelGGL(k_mp_dipole_field, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, pD, pP, a.phiInd, a.phiIndP, tD, tP, w, 1);      // T p, Ap, p.Ap
        hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, a, w, 2, 0.0, 0.0);          // mu += a p, r -= a Ap (a from the device sums)
        if (a.precond) hipLaunchKernelGGL(k_mp_precond, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, w, 0);      // z = M r, r.z
        hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, aClear, w, 3, mp->target_epsilon, 0.0);   // p = z + b p; the last block rolls the sums and forms the measure and the convergence word
        enqueued++;
        if (enqueued >= unchecked || enqueued == mp->max_iterations) { rc = readSums(); if (rc != 0) return rc; done = h[10] != 0.0; }
    }
    // nothing read yet (expected_iterations > 1 with max_iterations <= 0: the loop never ran): the first guess may already be converged
    if (unchecked != 0 && enqueued == 0) { rc = readSums(); if (rc != 0) return rc; done = h[10] != 0.0; }
    a.doneFlag = nullptr;
    const double epsilon = h[12];
    const int iterations = (int) h[11];
    if (mp->status != nullptr) { mp->status[0] = epsilon; mp->status[1] = iterations; }
    static const bool report = getenv("OPENMM_HIP_AMOEBA_DEBUG") != nullptr;
    if (report) fprintf(stderr, "amoeba solver: %d iterations (%d enqueued, first guess from %d earlier solutions), epsilon %.3g (target %.3g), preconditioner %d\n", iterations, enqueued, use, epsilon, mp->target_epsilon, a.precond);
    if (!done) return -1;
    if (tailEnqueued) { mainBesideFinal(); return 0; }          // (the last update, the history record and the final potentials are in the queues already)
    if (!noPolish) hipLaunchKernelGGL(k_mp_cg, dim3(cgBlocks), dim3(MP_CG_BLOCK), 0, st, a, w, 6, 0.0, 0.0);
    if (haveHistory && mp->history_store >= 0)
        hipLaunchKernelGGL(k_mp_history, dim3(blocks), dim3(MP_BLOCK), 0, st, a, mp->history, mp->history_slots, mp->history_store % mp->history_slots, 0, 1, coeff);
    // potentials of the converged dipoles (the force kernels read them).  After a folded solve the grids are still zero: the stage 7 that found
    // the convergence cleared them, and what was enqueued behind it has left at once (the transforms ran on zeros)
    if (finalOnSide) {
        // (round 5) on the solver's side stream: the caller runs the list pairs of the force kernel, which read no potential, beside this chain
        // and waits for event_b in front of the kernel that does (ommhip_amoeba_multipole_forces)
        hipStream_t st2 = (hipStream_t) mp->stream2;
        hipEventRecord((hipEvent_t) mp->event_a, st);
        mainBesideFinal();
        hipStreamWaitEvent(st2, (hipEvent_t) mp->event_a, 0);
        dipole_potentials(mp, a, a.indD, a.phiInd, a.indP, a.phiIndP, st2, false, 0, fold && enqueued > 0);
        hipEventRecord((hipEvent_t) mp->event_b, st2);
    }
    else dipole_potentials(mp, a, a.indD, a.phiInd, a.indP, a.phiIndP, st, false, 0, fold && enqueued > 0);
    return 0;
}
int solve_mutual(const ommhip_amoeba_multipole* mp, MpArgs a, hipStream_t st, bool finalOnSide = false, int* deferred = nullptr) { return solve_mutual(mp, a, st, finalOnSide, deferred, [] {}); }

// Extrapolated polarization: orders 1 .. K - 1 from the direct dipoles, the total dipoles, their potentials.  Leaves mu_d, mu_p (totals) in
// indD / indP, every order in mp->ext_dipoles and the field gradients of orders 0 .. K - 2 in mp->ext_gradients.
int solve_extrapolated(const ommhip_amoeba_multipole* mp, const MpArgs& a, hipStream_t st) {
    const int K = mp->extrapolation_orders;
    if (K < 1 || K > OMMHIP_AMOEBA_MAX_EXT_ORDERS || mp->ext_dipoles == nullptr || (K > 1 && mp->ext_gradients == nullptr) || mp->solver == nullptr || mp->phi_induced_p == nullptr) return 1;
    const int blocks = (a.n + MP_BLOCK - 1) / MP_BLOCK;
    const size_t n3 = 3 * (size_t) a.n, n6 = 6 * (size_t) a.n;
    double* fD = mp->solver + 6 * n3; double* fP = mp->solver + 7 * n3;
    ExtCoeff coeff;
    for (int k = 0; k < OMMHIP_AMOEBA_MAX_EXT_ORDERS; k++) {          // P_k = sum_(j >= k) c_j (AmoebaReferenceMultipoleForce.cpp:166-173)
        coeff.p[k] = 0.0;
        for (int j = k; j < K; j++) coeff.p[k] += mp->ext_coefficients[j];
    }
    hipLaunchKernelGGL(k_mp_ext_step, dim3(blocks), dim3(MP_BLOCK), 0, st, a, mp->ext_dipoles, 0, K, 0, (const double*) nullptr, (const double*) nullptr, coeff);
    for (int order = 1; order < K; order++) {
        dipole_potentials(mp, a, a.indD, a.phiInd, a.indP, a.phiIndP, st, false, 2);
        double* gD = mp->ext_gradients + (size_t) (order - 1) * 2 * n6;
        if (mp->mixed_precision) hipLaunchKernelGGL(k_mp_dipole_field_gradient<true>, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, a.indD, a.indP, a.phiInd, a.phiIndP, fD, fP, gD, gD + n6);
        else hipLaunchKernelGGL(k_mp_dipole_field_gradient<false>, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a, a.indD, a.indP, a.phiInd, a.phiIndP, fD, fP, gD, gD + n6);
        hipLaunchKernelGGL(k_mp_ext_step, dim3(blocks), dim3(MP_BLOCK), 0, st, a, mp->ext_dipoles, order, K, 1, fD, fP, coeff);
    }
    hipLaunchKernelGGL(k_mp_ext_step, dim3(blocks), dim3(MP_BLOCK), 0, st, a, mp->ext_dipoles, 0, K, 2, (const double*) nullptr, (const double*) nullptr, coeff);
    return 0;
}

}  // namespace

extern "C" int ommhip_amoeba_multipole_induce(const ommhip_amoeba_multipole* mp, const void* pos_d, const double box[6], void* stream) {
    MpArgs a;
    if (!make_args(mp, pos_d, box, a)) return 1;
    { const int rc = launch_induce(mp, a, box, (hipStream_t) stream); if (rc != 0) return rc; }
    if (a.mutual) { const int rc = solve_mutual(mp, a, (hipStream_t) stream); if (rc != 0) return rc; }
    else if (mp->extrapolation_orders > 0) { const int rc = solve_extrapolated(mp, a, (hipStream_t) stream); if (rc != 0) return rc; }
    return (int) hipGetLastError();
}

extern "C" int ommhip_amoeba_multipole_forces(const ommhip_amoeba_multipole* mp, const void* pos_d, const double box[6], const int* slot_of_atom_d, int padded_atoms,
                                              long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    MpArgs a;
    if (!make_args(mp, pos_d, box, a)) return 1;
    a.paddedAtoms = padded_atoms; a.includeEnergy = include_energy; a.energySlots = energy_slots;
    a.slotOfAtom = slot_of_atom_d; a.force = force_d; a.energyBuffer = energy_buffer_d;
    hipStream_t st = (hipStream_t) stream;
    const ommhip_pme* pme = (const ommhip_pme*) mp->pme;
    const int blocks = (a.n + MP_BLOCK - 1) / MP_BLOCK;
    static const bool checkAtOnce = getenv("OPENMM_HIP_AMOEBA_LIST_CHECK_AT_ONCE") != nullptr;       // A/B: the host waits for the list build at once
    int* const deferred = checkAtOnce ? nullptr : deferred_words();
    { const int rc = launch_induce(mp, a, box, st, true, deferred); if (rc != 0) return rc; }
    // Mutual polarization, mixed precision: the potentials of the converged dipoles (~100 us of small launches) are formed on the side stream while
    // k_mp_forces<true> -- the list pairs: no potential read -- runs on this one; k_mp_special<true>, which adds the reciprocal-space and self
    // terms, follows behind the wait and ADDS its torques to the ones the pair kernel stored (the other order before round 5).
    static const bool finalInline = getenv("OPENMM_HIP_AMOEBA_FINAL_INLINE") != nullptr;       // A/B
    const bool finalOnSide = a.mutual && mp->mixed_precision && !finalInline && mp->stream2 != nullptr && mp->event_a != nullptr && mp->event_b != nullptr && two_grid_launches(mp, a);
    bool pairForcesLaunched = false;
    if (a.mutual) {
        const int rc = solve_mutual(mp, a, st, finalOnSide, deferred, [&] {
            if (!mp->mixed_precision) return;
            MpArgs f = a; f.specialAdds = 1;
            hipLaunchKernelGGL(k_mp_forces<true>, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, f);
            pairForcesLaunched = true;
        });
        if (rc != 0) return rc;      // -1: not converged; -2: the lists (deferred check)
    }
    else {
        if (mp->extrapolation_orders > 0) { const int rc = solve_extrapolated(mp, a, st); if (rc != 0) return rc; }
        // reciprocal potential of the induced dipoles (mu_d + mu_p) / 2
        hipMemsetAsync(a.grid, 0, sizeof(float) * (size_t) a.nx * a.ny * a.nz, st);
        spread_induced(a, a.indD, 0.5, a.indP, 0.5, st);
        ommhip_pme_convolve(pme, st);
        hipLaunchKernelGGL(k_mp_potential<3>, dim3(spread_blocks(a)), dim3(256), 0, st, a, a.phiInd, (double*) nullptr);
    }
    if (deferred != nullptr && !a.mutual) {
        // no solver wait on the way: one here, before the first kernel that adds to the forces
        const hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) return (int) e;
        const int rc = deferred_lists_result(mp, deferred);
        if (rc != 0) return rc;
    }
    if (mp->mixed_precision && finalOnSide) {
        // k_mp_special<true> follows the potentials on the side stream -- beside the pair kernel too -- and leaves its torques in a vector of their
        // own (the solver's t_d: free by now); k_mp_torque_to_force, behind the wait, adds the two.  Forces and energies meet in atomics anyway.
        static const bool specialOnMain = getenv("OPENMM_HIP_AMOEBA_SPECIAL_ON_MAIN") != nullptr;       // A/B
        a.specialAdds = 1;
        if (!specialOnMain) {
            a.torque2 = mp->solver + 6 * 3 * (size_t) a.n;
            hipLaunchKernelGGL(k_mp_special<true>, dim3((unsigned) (((size_t) a.n * MP_SPLIT + MP_BLOCK - 1) / MP_BLOCK)), dim3(MP_BLOCK), 0, (hipStream_t) mp->stream2, a);
            hipEventRecord((hipEvent_t) mp->event_b, (hipStream_t) mp->stream2);
        }
        if (!pairForcesLaunched) hipLaunchKernelGGL(k_mp_forces<true>, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a);
        hipStreamWaitEvent(st, (hipEvent_t) mp->event_b, 0);
        if (specialOnMain) hipLaunchKernelGGL(k_mp_special<true>, dim3((unsigned) (((size_t) a.n * MP_SPLIT + MP_BLOCK - 1) / MP_BLOCK)), dim3(MP_BLOCK), 0, st, a);
    }
    else if (mp->mixed_precision) {
        hipLaunchKernelGGL(k_mp_special<true>, dim3((unsigned) (((size_t) a.n * MP_SPLIT + MP_BLOCK - 1) / MP_BLOCK)), dim3(MP_BLOCK), 0, st, a);
        hipLaunchKernelGGL(k_mp_forces<true>, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a);
    }
    else hipLaunchKernelGGL(k_mp_forces<false>, dim3(scan_blocks(a)), dim3(MP_BLOCK), 0, st, a);
    if (!a.mutual && mp->extrapolation_orders > 1) {
        ExtCoeff coeff;
        for (int k = 0; k < OMMHIP_AMOEBA_MAX_EXT_ORDERS; k++) { coeff.p[k] = 0.0; for (int j = k; j < mp->extrapolation_orders; j++) coeff.p[k] += mp->ext_coefficients[j]; }
        hipLaunchKernelGGL(k_mp_ext_forces, dim3(blocks), dim3(MP_BLOCK), 0, st, a, mp->ext_dipoles, mp->ext_gradients, mp->extrapolation_orders, coeff);
    }
    hipLaunchKernelGGL(k_mp_torque_to_force, dim3(blocks), dim3(MP_BLOCK), 0, st, a);
    return (int) hipGetLastError();
}
