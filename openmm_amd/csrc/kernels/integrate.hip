// Integrators (Verlet, Langevin, LangevinMiddle), constraint solvers (SETTLE, SHAKE clusters, CCMA)
// and kinetic energy, all in FP64 on the atom-ordered state arrays
//     pos  double4[N] (x,y,z,-)     vel double4[N] (vx,vy,vz,1/m)     xp double4[N] (trial positions).
//
// Replaces (behaviourally)
//   ReferenceVerletDynamics.cpp:76-119, ReferenceStochasticDynamics.cpp:89-194,
//   ReferenceLangevinMiddleDynamics.cpp:54-127, ReferenceSETTLEAlgorithm.cpp:54-244,
//   ReferenceCCMAAlgorithm.cpp:205-316, ReferenceKernels.cpp:146-176 (kinetic energy)
// reached through IntegrateVerletStepKernel / IntegrateLangevinStepKernel /
// IntegrateLangevinMiddleStepKernel / ApplyConstraintsKernel (olla/include/openmm/kernels.h:1033,1160,1193,220).
//
// MI355X note: FP64 vector throughput is half of FP32 on this chip, and this stage moves ~100 bytes
// per atom, so it is bandwidth/latency bound either way; keeping the whole integration state in
// double replaces the float+correction ("mixed") scheme GPU platforms normally need.
#include "common.h"
#include "../../../include/openmm_hip_kernels.h"

#include "rng.h"

using namespace omm;

namespace {

struct IntArgs {
    int numAtoms, paddedAtoms;
    double dt, vscale, fscale, noisescale;
    unsigned long long seed, step;
    double4* pos; double4* vel; double4* xp; double4* oldx;
    const omm_fixed* force;
    const int* slotOfAtom;
    int* freeze;              // neighbour-list state array or null (ommhip_integrator_state::freeze_state)
};

// True while the neighbour list of this step's forces had overflowed: the kernel must not touch the state.  `count`: this
// kernel opens a step, so one of its threads records the skipped step.
__device__ __forceinline__ bool frozen(const IntArgs& a, bool count) {
    if (a.freeze == nullptr || a.freeze[OMMHIP_NL_STATE_OVERFLOW] == 0) return false;
    if (count && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.freeze[OMMHIP_NL_STATE_FROZEN], 1);
    return true;
}

__device__ __forceinline__ double3 load_force(const IntArgs& a, int atom) {
    int s = a.slotOfAtom[atom];
    return make_double3(from_fixed(a.force[s]), from_fixed(a.force[s + a.paddedAtoms]), from_fixed(a.force[s + 2 * a.paddedAtoms]));
}

// ReferenceVerletDynamics.cpp:97-104
__global__ void k_verlet_part1(IntArgs a) {
    if (frozen(a, true)) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    double4 x = a.pos[i], v = a.vel[i];
    if (v.w != 0.0) {
        double3 f = load_force(a, i);
        v.x += v.w * f.x * a.dt; v.y += v.w * f.y * a.dt; v.z += v.w * f.z * a.dt;
        a.vel[i] = v;
        x.x += v.x * a.dt; x.y += v.y * a.dt; x.z += v.z * a.dt;
    }
    a.xp[i] = x;
}
// ReferenceVerletDynamics.cpp:109-116 (also the last stage of ReferenceStochasticDynamics.cpp:138-146)
__global__ void k_finish_positions(IntArgs a) {
    if (frozen(a, false)) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    double4 x = a.pos[i], v = a.vel[i], xp = a.xp[i];
    if (v.w != 0.0) {
        double inv = 1.0 / a.dt;
        v.x = inv * (xp.x - x.x); v.y = inv * (xp.y - x.y); v.z = inv * (xp.z - x.z);
        a.vel[i] = v;
        a.pos[i] = make_double4(xp.x, xp.y, xp.z, x.w);
    }
}
// ReferenceStochasticDynamics.cpp:89-136
__global__ void k_langevin_part1(IntArgs a) {
    if (frozen(a, true)) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    double4 x = a.pos[i], v = a.vel[i];
    if (v.w != 0.0) {
        double3 f = load_force(a, i);
        double3 g = gaussian3((unsigned) i, a.step, a.seed);
        double sq = sqrt(v.w);
        v.x = a.vscale * v.x + a.fscale * v.w * f.x + a.noisescale * sq * g.x;
        v.y = a.vscale * v.y + a.fscale * v.w * f.y + a.noisescale * sq * g.y;
        v.z = a.vscale * v.z + a.fscale * v.w * f.z + a.noisescale * sq * g.z;
        a.vel[i] = v;
        x.x += v.x * a.dt; x.y += v.y * a.dt; x.z += v.z * a.dt;
    }
    a.xp[i] = x;
}
// ReferenceLangevinMiddleDynamics.cpp:54-58
__global__ void k_lmiddle_part1(IntArgs a) {
    if (frozen(a, true)) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    double4 v = a.vel[i];
    if (v.w != 0.0) {
        double3 f = load_force(a, i);
        v.x += a.dt * v.w * f.x; v.y += a.dt * v.w * f.y; v.z += a.dt * v.w * f.z;
        a.vel[i] = v;
    }
}
// ReferenceLangevinMiddleDynamics.cpp:60-80   (noisescale = sqrt(kT (1-vscale^2)))
__global__ void k_lmiddle_part2(IntArgs a) {
    if (frozen(a, false)) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    double4 x = a.pos[i], v = a.vel[i];
    if (v.w != 0.0) {
        const double h = 0.5 * a.dt;
        x.x += v.x * h; x.y += v.y * h; x.z += v.z * h;
        double3 g = gaussian3((unsigned) i, a.step, a.seed);
        double sq = sqrt(v.w);
        v.x = a.vscale * v.x + a.noisescale * sq * g.x;
        v.y = a.vscale * v.y + a.noisescale * sq * g.y;
        v.z = a.vscale * v.z + a.noisescale * sq * g.z;
        a.vel[i] = v;
        x.x += v.x * h; x.y += v.y * h; x.z += v.z * h;
    }
    a.xp[i] = x;
    a.oldx[i] = x;
}
// ReferenceLangevinMiddleDynamics.cpp:82-90
__global__ void k_lmiddle_part3(IntArgs a) {
    if (frozen(a, false)) return;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    double4 x = a.pos[i], v = a.vel[i], xp = a.xp[i], ox = a.oldx[i];
    if (v.w != 0.0) {
        double inv = 1.0 / a.dt;
        v.x += (xp.x - ox.x) * inv; v.y += (xp.y - ox.y) * inv; v.z += (xp.z - ox.z) * inv;
        a.vel[i] = v;
        a.pos[i] = make_double4(xp.x, xp.y, xp.z, x.w);
    }
}
// out = vel + force*(shift/m)   (ReferenceKernels.cpp:155-160)
__global__ void k_shifted_velocities(IntArgs a, double shift, double4* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.numAtoms) return;
    double4 v = a.vel[i];
    if (v.w != 0.0 && shift != 0.0) {
        double3 f = load_force(a, i);
        v.x += f.x * shift * v.w; v.y += f.y * shift * v.w; v.z += f.z * shift * v.w;
    }
    out[i] = v;
}
// Kinetic energy in two launches: up to KE_PARTIALS workgroups each sum a strided share of the range into scratch[block]
// (fixed assignment and summation order: the result is reproducible), one workgroup adds the partials.  The range is
// either atoms [first, end) or -- atomOfSlot given -- the atoms in the slots [first, end) (a rank's own atoms in a decomposed run).
#define KE_PARTIALS 1024
__global__ __launch_bounds__(256) void k_kinetic_energy_partial(const double4* __restrict__ vel, const int* __restrict__ atomOfSlot, int first, int end,
                                                                double* __restrict__ scratch) {
    __shared__ double part[4];
    double e = 0;
    for (int i = first + blockIdx.x * 256 + threadIdx.x; i < end; i += gridDim.x * 256) {
        const int atom = atomOfSlot != nullptr ? atomOfSlot[i] : i;
        if (atom < 0) continue;
        double4 v = vel[atom];
        if (v.w != 0.0) e += (v.x * v.x + v.y * v.y + v.z * v.z) / v.w;
    }
    e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void k_kinetic_energy_final(const double* __restrict__ scratch, int n, double* __restrict__ result) {
    __shared__ double part[4];
    double e = 0;
    for (int i = threadIdx.x; i < n; i += 256) e += scratch[i];
    e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) result[0] = 0.5 * (part[0] + part[1] + part[2] + part[3]);
}

// ================================================================================================
// SETTLE (Miyamoto & Kollman 1992) -- analytic rigid three-site reset, general masses.
// cluster: atoms (a0 = apex, a1, a2), params (apex-leg distance d01 = d02, base distance d12).
// ================================================================================================
struct SettleArgs {
    int numClusters;
    const int4* atoms;        // (a0, a1, a2, unused)
    const double2* dist;      // (d0x, d12)
    const double4* pos;       // positions before the step (constrained)
    double4* xp;              // trial positions, corrected in place   (position version)
    double4* vel;             // velocities, corrected in place         (velocity version; w = 1/m)
    const double4* velMass;   // array holding 1/m in .w
};

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ V3 xyz(double4 p) { return v3(p.x, p.y, p.z); }

// Same algebra as ReferenceSETTLEAlgorithm.cpp:54-195, written with vectors.  Register form: p = constrained
// positions before the step, q = trial positions (corrected in place), m = masses, (d01, d12) = leg and base lengths.
__device__ __forceinline__ void settle_positions_regs(const V3 p0, const V3 p1, const V3 p2, V3& q0, V3& q1, V3& q2,
                                                      const double m0, const double m1, const double m2, const double d01, const double d12) {
    const double2 dd = make_double2(d01, d12);
    // everything relative to the old apex position
    const V3 b0 = p1 - p0, c0 = p2 - p0;
    const V3 d0 = q0 - p0;                        // trial apex
    const V3 d1 = b0 + (q1 - p1);                 // trial leg atoms
    const V3 d2 = c0 + (q2 - p2);
    const double invM = 1.0 / (m0 + m1 + m2);
    const V3 com = (d0 * m0 + d1 * m1 + d2 * m2) * invM;
    const V3 a1 = d0 - com, b1 = d1 - com, c1 = d2 - com;
    // orthonormal frame: z = normal of the old triangle, x = a1 x z, y = z x x
    V3 ez = cross(b0, c0);
    V3 ex = cross(a1, ez);
    V3 ey = cross(ez, ex);
    ex = ex * (1.0 / sqrt(dot(ex, ex)));
    ey = ey * (1.0 / sqrt(dot(ey, ey)));
    ez = ez * (1.0 / sqrt(dot(ez, ez)));
    const double xb0 = dot(ex, b0), yb0 = dot(ey, b0), xc0 = dot(ex, c0), yc0 = dot(ey, c0);
    const double za1 = dot(ez, a1);
    const double xb1 = dot(ex, b1), yb1 = dot(ey, b1), zb1 = dot(ez, b1);
    const double xc1 = dot(ex, c1), yc1 = dot(ey, c1), zc1 = dot(ez, c1);
    // canonical triangle
    const double rc = 0.5 * dd.y;
    double rb = sqrt(dd.x * dd.x - rc * rc);
    const double ra = rb * (m1 + m2) * invM;
    rb -= ra;
    const double sinphi = za1 / ra;
    const double cosphi = sqrt(1.0 - sinphi * sinphi);
    const double sinpsi = (zb1 - zc1) / (2.0 * rc * cosphi);
    const double cospsi = sqrt(1.0 - sinpsi * sinpsi);
    const double ya2 = ra * cosphi;
    double xb2 = -rc * cospsi;
    const double yb2 = -rb * cosphi - rc * sinpsi * sinphi;
    const double yc2 = -rb * cosphi + rc * sinpsi * sinphi;
    const double xb22 = xb2 * xb2;
    const double hh2 = 4.0 * xb22 + (yb2 - yc2) * (yb2 - yc2) + (zb1 - zc1) * (zb1 - zc1);
    const double deltx = 2.0 * xb2 + sqrt(4.0 * xb22 - hh2 + dd.y * dd.y);
    xb2 -= 0.5 * deltx;
    // rotation about z
    const double alpha = xb2 * (xb0 - xc0) + yb0 * yb2 + yc0 * yc2;
    const double beta = xb2 * (yc0 - yb0) + xb0 * yb2 + xc0 * yc2;
    const double gamma = xb0 * yb1 - xb1 * yb0 + xc0 * yc1 - xc1 * yc0;
    const double al2be2 = alpha * alpha + beta * beta;
    const double sintheta = (alpha * gamma - beta * sqrt(al2be2 - gamma * gamma)) / al2be2;
    const double costheta = sqrt(1.0 - sintheta * sintheta);
    const V3 a3 = ex * (-ya2 * sintheta) + ey * (ya2 * costheta) + ez * za1;
    const V3 b3 = ex * (xb2 * costheta - yb2 * sintheta) + ey * (xb2 * sintheta + yb2 * costheta) + ez * zb1;
    const V3 c3 = ex * (-xb2 * costheta - yc2 * sintheta) + ey * (-xb2 * sintheta + yc2 * costheta) + ez * zc1;
    q0 = p0 + com + a3;
    q1 = p1 + (com + b3 - b0);
    q2 = p2 + (com + c3 - c0);
}
__device__ __forceinline__ void settle_positions_cluster(const SettleArgs& a, int c) {
    const int4 at = a.atoms[c];
    const double2 dd = a.dist[c];
    const double4 q0 = a.xp[at.x], q1 = a.xp[at.y], q2 = a.xp[at.z];
    V3 n0 = xyz(q0), n1 = xyz(q1), n2 = xyz(q2);
    settle_positions_regs(xyz(a.pos[at.x]), xyz(a.pos[at.y]), xyz(a.pos[at.z]), n0, n1, n2,
                          1.0 / a.velMass[at.x].w, 1.0 / a.velMass[at.y].w, 1.0 / a.velMass[at.z].w, dd.x, dd.y);
    a.xp[at.x] = make_double4(n0.x, n0.y, n0.z, q0.w);
    a.xp[at.y] = make_double4(n1.x, n1.y, n1.z, q1.w);
    a.xp[at.z] = make_double4(n2.x, n2.y, n2.z, q2.w);
}

// ReferenceSETTLEAlgorithm.cpp:197-244 (unequal-mass velocity solve)
__device__ __forceinline__ void settle_velocities_regs(const V3 p0, const V3 p1, const V3 p2, V3& v0, V3& v1, V3& v2,
                                                       const double iA, const double iB, const double iC) {
    const double mA = 1.0 / iA, mB = 1.0 / iB, mC = 1.0 / iC;
    V3 eAB = p1 - p0, eBC = p2 - p1, eCA = p0 - p2;
    eAB = eAB * (1.0 / sqrt(dot(eAB, eAB)));
    eBC = eBC * (1.0 / sqrt(dot(eBC, eBC)));
    eCA = eCA * (1.0 / sqrt(dot(eCA, eCA)));
    const double vAB = dot(v1 - v0, eAB), vBC = dot(v2 - v1, eBC), vCA = dot(v0 - v2, eCA);
    const double cA = -dot(eAB, eCA), cB = -dot(eAB, eBC), cC = -dot(eBC, eCA);
    const double s2A = 1 - cA * cA, s2B = 1 - cB * cB, s2C = 1 - cC * cC;
    const double mABCinv = 1 / (mA * mB * mC);
    const double denom = (((s2A * mB + s2B * mA) * mC + (s2A * mB * mB + 2 * (cA * cB * cC + 1) * mA * mB + s2B * mA * mA)) * mC + s2C * mA * mB * (mA + mB)) * mABCinv;
    const double tab = ((cB * cC * mA - cA * mB - cA * mC) * vCA + (cA * cC * mB - cB * mC - cB * mA) * vBC + (s2C * mA * mA * mB * mB * mABCinv + (mA + mB + mC)) * vAB) / denom;
    const double tbc = ((cA * cB * mC - cC * mB - cC * mA) * vCA + (s2A * mB * mB * mC * mC * mABCinv + (mA + mB + mC)) * vBC + (cA * cC * mB - cB * mA - cB * mC) * vAB) / denom;
    const double tca = ((s2B * mA * mA * mC * mC * mABCinv + (mA + mB + mC)) * vCA + (cA * cB * mC - cC * mB - cC * mA) * vBC + (cB * cC * mA - cA * mB - cA * mC) * vAB) / denom;
    v0 = v0 + (eAB * tab - eCA * tca) * iA;
    v1 = v1 + (eBC * tbc - eAB * tab) * iB;
    v2 = v2 + (eCA * tca - eBC * tbc) * iC;
}
__device__ __forceinline__ void settle_velocities_cluster(const SettleArgs& a, int c) {
    const int4 at = a.atoms[c];
    const double4 w0 = a.vel[at.x], w1 = a.vel[at.y], w2 = a.vel[at.z];
    V3 v0 = xyz(w0), v1 = xyz(w1), v2 = xyz(w2);
    settle_velocities_regs(xyz(a.pos[at.x]), xyz(a.pos[at.y]), xyz(a.pos[at.z]), v0, v1, v2, a.velMass[at.x].w, a.velMass[at.y].w, a.velMass[at.z].w);
    a.vel[at.x] = make_double4(v0.x, v0.y, v0.z, w0.w);
    a.vel[at.y] = make_double4(v1.x, v1.y, v1.z, w1.w);
    a.vel[at.z] = make_double4(v2.x, v2.y, v2.z, w2.w);
}

// ================================================================================================
// SHAKE clusters: one central atom with 1..3 satellites that have no other constraint.  One thread
// iterates its cluster in registers until the Reference convergence criterion
// (ReferenceCCMAAlgorithm.cpp:246-275) holds; no host round trip.
// ================================================================================================
struct ShakeArgs {
    int numClusters, maxIterations;
    double tol;
    const int4* atoms;        // (center, s1, s2, s3), unused = -1
    const double4* dist;      // (d1, d2, d3, unused)
    const double4* pos;
    double4* target;          // xp (positions) or vel (velocities)
    const double4* velMass;
};

// Register form: r[k] = x_centre - x_satellite k (constrained positions before the step), t0 / t[k] = the vectors being
// corrected (trial positions or velocities), iw = inverse masses, n = number of satellites.
template <bool VELOCITIES>
__device__ __forceinline__ void shake_regs(const V3 (&r)[3], V3& t0, V3 (&t)[3], const double iw0, const double (&iw)[3],
                                           const double (&dist)[3], const int n, const double tol, const int maxIterations) {
    double rr[3];
#pragma unroll
    for (int k = 0; k < 3; k++) rr[k] = k < n ? dot(r[k], r[k]) : 1.0;
    const double lowerTol = 1 - 2 * tol + tol * tol, upperTol = 1 + 2 * tol + tol * tol;
    for (int iter = 0; iter < maxIterations; iter++) {
        bool converged = true;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k < n) {
                const double reduced = 0.5 / (iw0 + iw[k]);
                V3 rp = t0 - t[k];
                double delta;
                if (VELOCITIES) {
                    delta = -2.0 * reduced * dot(rp, r[k]) / rr[k];
                    if (fabs(delta) > tol) converged = false; else delta = 0.0;
                }
                else {
                    const double rp2 = dot(rp, rp), d2 = dist[k] * dist[k];
                    if (rp2 >= lowerTol * d2 && rp2 <= upperTol * d2) delta = 0.0;
                    else { delta = reduced * (d2 - rp2) / dot(rp, r[k]); converged = false; }
                }
                t0 = t0 + r[k] * (delta * iw0);
                t[k] = t[k] - r[k] * (delta * iw[k]);
            }
        }
        if (converged) break;
    }
}

template <bool VELOCITIES>
__device__ __forceinline__ void shake_cluster(const ShakeArgs& a, int c) {
    const int4 at = a.atoms[c];
    const double4 dd = a.dist[c];
    const int sat[3] = {at.y, at.z, at.w};
    const double dist[3] = {dd.x, dd.y, dd.z};
    const V3 x0 = xyz(a.pos[at.x]);
    const double iw0 = a.velMass[at.x].w;
    double4 t0w = a.target[at.x];
    V3 t0 = xyz(t0w);
    V3 r[3], t[3];
    double iw[3], tw[3];
    int n = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        r[k] = v3(0, 0, 0); t[k] = v3(0, 0, 0); iw[k] = 0; tw[k] = 0;
        if (sat[k] >= 0) {
            r[k] = x0 - xyz(a.pos[sat[k]]);
            double4 tt = a.target[sat[k]];
            t[k] = xyz(tt); tw[k] = tt.w;
            iw[k] = a.velMass[sat[k]].w;
            n = k + 1;
        }
    }
    shake_regs<VELOCITIES>(r, t0, t, iw0, iw, dist, n, a.tol, a.maxIterations);
    a.target[at.x] = make_double4(t0.x, t0.y, t0.z, t0w.w);
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (k < n) a.target[sat[k]] = make_double4(t[k].x, t[k].y, t[k].z, tw[k]);
}

// ================================================================================================
// Fused step: one thread integrates one *unit* -- a SETTLE water, a SHAKE cluster or a single unconstrained
// atom -- through the whole update (force kick, velocity constraints, drift + thermostat, position constraints,
// velocity correction), entirely in registers.  Replaces 5 launches (7 with the CM-motion remover, whose
// momentum sum for the next step is accumulated here).  Same arithmetic, in the same order, as the staged kernels.
//   KIND 0: ReferenceVerletDynamics.cpp:76-119   1: ReferenceStochasticDynamics.cpp:89-194
//   KIND 2: ReferenceLangevinMiddleDynamics.cpp:54-127
// ================================================================================================
struct UnitArgs {
    int numUnits, maxIterations, removeCm;
    double tol, invTotalMass;
    const int4* atoms;        // (a0, a1, a2, a3), unused = -1; SETTLE: a0 = apex; SHAKE: a0 = centre
    const double4* dist;      // SETTLE (d01, d12, -, 1); SHAKE (d1, d2, d3, 2); free atom (-, -, -, 0)
    double* cm;               // [0..2] momentum after the previous fused step, [3] block counter (as int), [4 + 4*block] partials
    // domain decomposition (posWire != null): new positions also go to the all-gather buffer as fixed-point box fractions, and the
    // momentum lives in two reserved records per rank of that buffer ("trailer": 32 bytes = three doubles), so the CM velocity
    // needs no collective of its own
    uint4* posWire;
    double invBox[3];         // 1 / ax, 1 / by, 1 / cz
    double skew[3];           // bx, cx, cy
    int ranks, rank, slotsPerRank, trailerSlot;
    const int* ddFlags;       // halo mode: [1] = an owned atom is near (1) or close to the end of (3) the drift margin -> fourth double of the trailer (every rank sees it one step later)
};

// SMALL: no unit is a SHAKE cluster (only SETTLE waters and free atoms: every unit has at most three atoms) -- the state of a fourth
// atom and the SHAKE iteration are compiled out, which is worth registers (two waves per SIMD otherwise) in a latency-bound kernel.
template <int KIND, bool SMALL>
__global__ __launch_bounds__(128) void k_step_units(IntArgs a, UnitArgs u) {
    constexpr int MAXA = SMALL ? 3 : 4;
    if (frozen(a, true)) return;
    const int c = blockIdx.x * 128 + threadIdx.x;
    V3 mom = v3(0, 0, 0);
    if (c < u.numUnits) {
        const int4 at = u.atoms[c];
        const double4 dd = u.dist[c];
        const int kind = (int) dd.w;
        const int ids[4] = {at.x, at.y, at.z, SMALL ? -1 : at.w};
        V3 x[MAXA], v[MAXA], xn[MAXA], ox[MAXA];
        double w[MAXA], xw[MAXA];
        V3 cmv = v3(0, 0, 0);
        if (u.removeCm) {
            if (u.posWire != nullptr) {
                // total momentum = the ranks' trailers summed in rank order (the same bits on every rank)
                double mx = 0, my = 0, mz = 0;
                for (int r = 0; r < u.ranks; r++) { const double4 m = *(const double4*) (u.posWire + (size_t) r * u.slotsPerRank + u.trailerSlot); mx += m.x; my += m.y; mz += m.z; }
                cmv = v3(mx * u.invTotalMass, my * u.invTotalMass, mz * u.invTotalMass);
            }
            else cmv = v3(u.cm[0] * u.invTotalMass, u.cm[1] * u.invTotalMass, u.cm[2] * u.invTotalMass);
        }
        const double h = 0.5 * a.dt;
#pragma unroll
        for (int k = 0; k < MAXA; k++) {
            x[k] = v3(0, 0, 0); v[k] = v3(0, 0, 0); w[k] = 0; xw[k] = 0;
            if (ids[k] >= 0) {
                const double4 p = a.pos[ids[k]], vv = a.vel[ids[k]];
                x[k] = xyz(p); xw[k] = p.w; v[k] = xyz(vv); w[k] = vv.w;
                if (w[k] != 0.0) {
                    v[k] = v[k] - cmv;                       // CMMotionRemover of this step (ReferenceKernels.cpp:2705-2740)
                    const double3 f = load_force(a, ids[k]);
                    if (KIND == 0 || KIND == 2) { v[k].x += a.dt * w[k] * f.x; v[k].y += a.dt * w[k] * f.y; v[k].z += a.dt * w[k] * f.z; }
                    if (KIND == 1) {
                        const double3 g = gaussian3((unsigned) ids[k], a.step, a.seed);
                        const double sq = sqrt(w[k]);
                        v[k].x = a.vscale * v[k].x + a.fscale * w[k] * f.x + a.noisescale * sq * g.x;
                        v[k].y = a.vscale * v[k].y + a.fscale * w[k] * f.y + a.noisescale * sq * g.y;
                        v[k].z = a.vscale * v[k].z + a.fscale * w[k] * f.z + a.noisescale * sq * g.z;
                    }
                }
            }
            xn[k] = x[k]; ox[k] = x[k];
        }
        V3 r[3];
        double iws[3] = {w[1], w[2], SMALL ? 0.0 : w[MAXA - 1]};
        const double dist[3] = {dd.x, dd.y, dd.z};
        int n = 0;
        if (!SMALL && kind == 2) {
#pragma unroll
            for (int k = 0; k < MAXA - 1; k++) { r[k] = x[0] - x[k + 1]; if (ids[k + 1] >= 0) n = k + 1; }
        }
        if (KIND == 2) {
            // velocity constraints on the kicked velocities (ReferenceLangevinMiddleDynamics.cpp:104)
            if (kind == 1) settle_velocities_regs(x[0], x[1], x[2], v[0], v[1], v[2], w[0], w[1], w[2]);
            else if (!SMALL && kind == 2) { V3 t[3] = {v[1], v[2], v[MAXA - 1]}; shake_regs<true>(r, v[0], t, w[0], iws, dist, n, u.tol, u.maxIterations); v[1] = t[0]; v[2] = t[1]; v[MAXA - 1] = t[2]; }
        }
#pragma unroll
        for (int k = 0; k < MAXA; k++) {
            if (ids[k] >= 0 && w[k] != 0.0) {
                if (KIND == 2) {
                    xn[k] = x[k] + v[k] * h;
                    const double3 g = gaussian3((unsigned) ids[k], a.step, a.seed);
                    const double sq = sqrt(w[k]);
                    v[k].x = a.vscale * v[k].x + a.noisescale * sq * g.x;
                    v[k].y = a.vscale * v[k].y + a.noisescale * sq * g.y;
                    v[k].z = a.vscale * v[k].z + a.noisescale * sq * g.z;
                    xn[k] = xn[k] + v[k] * h;
                    ox[k] = xn[k];
                }
                else xn[k] = x[k] + v[k] * a.dt;
            }
        }
        // position constraints against the positions at the start of the step
        if (kind == 1) settle_positions_regs(x[0], x[1], x[2], xn[0], xn[1], xn[2], 1.0 / w[0], 1.0 / w[1], 1.0 / w[2], dd.x, dd.y);
        else if (!SMALL && kind == 2) { V3 t[3] = {xn[1], xn[2], xn[MAXA - 1]}; shake_regs<false>(r, xn[0], t, w[0], iws, dist, n, u.tol, u.maxIterations); xn[1] = t[0]; xn[2] = t[1]; xn[MAXA - 1] = t[2]; }
        const double inv = 1.0 / a.dt;
#pragma unroll
        for (int k = 0; k < MAXA; k++) {
            if (ids[k] >= 0 && w[k] != 0.0) {
                if (KIND == 2) v[k] = v[k] + (xn[k] - ox[k]) * inv;
                else v[k] = (xn[k] - x[k]) * inv;
                a.vel[ids[k]] = make_double4(v[k].x, v[k].y, v[k].z, w[k]);
                a.pos[ids[k]] = make_double4(xn[k].x, xn[k].y, xn[k].z, xw[k]);
                if (u.posWire != nullptr) {
                    // fraction of the box edge in [0, 1) as 32-bit fixed point (the wrap into the box is the conversion's modulo)
                    // (coefficients of the box vectors c, b, a in turn; a rectangular box has skew = 0 and these are x / ax, y / by, z / cz)
                    const double fz0 = xn[k].z * u.invBox[2], fy0 = (xn[k].y - fz0 * u.skew[2]) * u.invBox[1];
                    double fx = (xn[k].x - fy0 * u.skew[0] - fz0 * u.skew[1]) * u.invBox[0], fy = fy0, fz = fz0;
                    fx -= floor(fx); fy -= floor(fy); fz -= floor(fz);
                    u.posWire[a.slotOfAtom[ids[k]]] = make_uint4((unsigned) (unsigned long long) (fx * 4294967296.0), (unsigned) (unsigned long long) (fy * 4294967296.0),
                                                               (unsigned) (unsigned long long) (fz * 4294967296.0), 0u);
                }
                mom = mom + v[k] * (1.0 / w[k]);
            }
        }
    }
    // ---- momentum of the new velocities, for the CMMotionRemover of the next step
    if (u.cm == nullptr) return;
    __shared__ double part[2][3];
    __shared__ bool last;
    mom.x = wave_sum(mom.x); mom.y = wave_sum(mom.y); mom.z = wave_sum(mom.z);
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = mom.x; part[threadIdx.x >> 6][1] = mom.y; part[threadIdx.x >> 6][2] = mom.z; }
    __syncthreads();
    int* counter = (int*) (u.cm + 3);
    // Hand-over of the block partials to the last block.  OMM_CM_TAIL_ATOMICS = 1: the three partial sums are written with
    // returning device-scope atomics (they complete at the memory side before the counter is touched) and read back the same
    // way -- no __threadfence(), whose L2 write-back (buffer_wbl2) would flush everything this kernel has just written on the
    // XCD, once per block, on the critical path of the step's last launch.  0: plain stores + fence (the form until session 2).
#ifndef OMM_CM_TAIL_ATOMICS
#define OMM_CM_TAIL_ATOMICS 1
#endif
    if (threadIdx.x == 0) {
        double* out = u.cm + 4 + 4 * blockIdx.x;
        const double sx = part[0][0] + part[1][0], sy = part[0][1] + part[1][1], sz = part[0][2] + part[1][2];
        if (OMM_CM_TAIL_ATOMICS) {
            unsigned long long seen = atomicExch((unsigned long long*) &out[0], (unsigned long long) __double_as_longlong(sx));
            seen |= atomicExch((unsigned long long*) &out[1], (unsigned long long) __double_as_longlong(sy));
            seen |= atomicExch((unsigned long long*) &out[2], (unsigned long long) __double_as_longlong(sz));
            // The returned values are consumed before the counter is touched: the three exchanges have completed by then.  Hardware
            // assumption: a returning device-scope atomic has been performed at L2 when its value comes back.  The "memory" clobber is a
            // compiler-only barrier (no buffer_wbl2): it keeps the counter's atomicAdd below from being moved above the exchanges.
#ifdef OMMHIP_EMU
            (void) seen;
#else
            asm volatile("" :: "v"(seen) : "memory");
#endif
            last = atomicAdd(counter, 1) == (int) gridDim.x - 1;
        }
        else {
            out[0] = sx; out[1] = sy; out[2] = sz;
            __threadfence();
            last = atomicAdd(counter, 1) == (int) gridDim.x - 1;
        }
    }
    __syncthreads();
    if (last && threadIdx.x < 64) {
        // fixed summation order -> the same bits whatever the block scheduling
        double sx = 0, sy = 0, sz = 0;
        if (OMM_CM_TAIL_ATOMICS) {
            // eight blocks per lane are requested before the first is added (2 567 blocks at a million atoms: 5 round trips per
            // lane instead of 40); the order of the additions is the one of a plain loop over b
            constexpr int BATCH = 8;
            for (int b0 = threadIdx.x; b0 < (int) gridDim.x; b0 += 64 * BATCH) {
                unsigned long long r[BATCH][3];
#pragma unroll
                for (int k = 0; k < BATCH; k++) {
                    const int b = b0 + 64 * k;
                    r[k][0] = r[k][1] = r[k][2] = 0ull;          // the bits of +0.0
                    if (b < (int) gridDim.x) {
                        unsigned long long* in = (unsigned long long*) (u.cm + 4 + 4 * b);
                        // a real read-modify-write (an add of 0 would be folded into a load, which the XCD's L2 may serve from a stale line)
                        r[k][0] = atomicExch(&in[0], 0ull); r[k][1] = atomicExch(&in[1], 0ull); r[k][2] = atomicExch(&in[2], 0ull);
                    }
                }
#pragma unroll
                for (int k = 0; k < BATCH; k++) {
                    sx += __longlong_as_double((long long) r[k][0]); sy += __longlong_as_double((long long) r[k][1]); sz += __longlong_as_double((long long) r[k][2]);
                }
            }
        }
        else {
            for (int b = threadIdx.x; b < (int) gridDim.x; b += 64) {
                const volatile double* in = u.cm + 4 + 4 * b;
                sx += in[0]; sy += in[1]; sz += in[2];
            }
        }
        sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
        if (threadIdx.x == 0) {
            u.cm[0] = sx; u.cm[1] = sy; u.cm[2] = sz; *counter = 0;
            if (u.posWire != nullptr) *(double4*) (u.posWire + (size_t) u.rank * u.slotsPerRank + u.trailerSlot) = make_double4(sx, sy, sz, u.ddFlags != nullptr ? (double) u.ddFlags[1] : 0.0);
        }
    }
}

// One launch for both cluster kinds: workgroups [0, shakeBlocks) iterate SHAKE clusters, the rest SETTLE waters.
template <bool VELOCITIES>
__global__ __launch_bounds__(128) void k_constrain_clusters(ShakeArgs sh, SettleArgs se, int shakeBlocks) {
    if ((int) blockIdx.x < shakeBlocks) {
        const int c = blockIdx.x * 128 + threadIdx.x;
        if (c < sh.numClusters) shake_cluster<VELOCITIES>(sh, c);
    }
    else {
        const int c = ((int) blockIdx.x - shakeBlocks) * 128 + threadIdx.x;
        if (c < se.numClusters) {
            if (VELOCITIES) settle_velocities_cluster(se, c);
            else settle_positions_cluster(se, c);
        }
    }
}

// ================================================================================================
// CCMA (general constraints) -- ReferenceCCMAAlgorithm.cpp:205-316.  The host drives the iteration
// and reads the converged count; only systems with constraints outside SETTLE/SHAKE take this path.
// ================================================================================================
struct CcmaArgs {
    int numConstraints, velocities;
    double tol;
    const int2* atoms;
    const double* dist;
    const double4* pos;
    double4* target;
    const double4* velMass;
    double* delta;            // [numConstraints]
    double* delta2;
    const int* rowStart; const int* col; const double* value;
    int* converged;           // [0] converged constraints of the iteration in flight, [1] blocks done, [2] all converged (sticky), [3] iterations run
    int resident;             // 1: the iteration kernels look at [2] and leave once it is set (batches of iterations without host round trips)
};

__global__ __launch_bounds__(128) void k_ccma_delta(CcmaArgs a) {
    if (a.resident && a.converged[2] != 0) return;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (c < a.numConstraints) {
    const int2 at = a.atoms[c];
    const V3 r = xyz(a.pos[at.x]) - xyz(a.pos[at.y]);
    const V3 rp = xyz(a.target[at.x]) - xyz(a.target[at.y]);
    const double reduced = 0.5 / (a.velMass[at.x].w + a.velMass[at.y].w);
    double delta;
    if (a.velocities) {
        delta = -2.0 * reduced * dot(rp, r) / dot(r, r);
        ok = fabs(delta) <= a.tol;
    }
    else {
        const double rp2 = dot(rp, rp), d2 = a.dist[c] * a.dist[c];
        delta = reduced * (d2 - rp2) / dot(rp, r);
        const double lowerTol = 1 - 2 * a.tol + a.tol * a.tol, upperTol = 1 + 2 * a.tol + a.tol * a.tol;
        ok = rp2 >= lowerTol * d2 && rp2 <= upperTol * d2;
    }
    a.delta[c] = delta;
    }
    const int waveOk = __popcll(__ballot(ok));
    if (lane_id() == 0 && waveOk > 0) atomicAdd(a.converged, waveOk);
    if (!a.resident) return;
    // device-resident loop: the last workgroup of the iteration decides whether everything has converged
    __shared__ bool last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        last = atomicAdd(&a.converged[1], 1) == (int) gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        if (atomicAdd(&a.converged[0], 0) == a.numConstraints) a.converged[2] = 1;
        a.converged[0] = 0; a.converged[1] = 0;
        a.converged[3] += 1;
    }
}
__global__ void k_ccma_multiply(CcmaArgs a) {
    if (a.resident && a.converged[2] != 0) return;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.numConstraints) return;
    double sum = 0;
    for (int e = a.rowStart[c]; e < a.rowStart[c + 1]; e++) sum += a.value[e] * a.delta[a.col[e]];
    a.delta2[c] = sum;
}
__global__ void k_ccma_update(CcmaArgs a) {
    if (a.resident && a.converged[2] != 0) return;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= a.numConstraints) return;
    const int2 at = a.atoms[c];
    const V3 r = xyz(a.pos[at.x]) - xyz(a.pos[at.y]);
    const double d = a.delta2[c];
    const double wi = a.velMass[at.x].w * d, wj = a.velMass[at.y].w * d;
    double* ti = (double*) &a.target[at.x];
    double* tj = (double*) &a.target[at.y];
    atomicAdd(ti, r.x * wi); atomicAdd(ti + 1, r.y * wi); atomicAdd(ti + 2, r.z * wi);
    atomicAdd(tj, -r.x * wj); atomicAdd(tj + 1, -r.y * wj); atomicAdd(tj + 2, -r.z * wj);
}

IntArgs make_int_args(const ommhip_integrator_state* s) {
    IntArgs a;
    a.numAtoms = s->num_atoms; a.paddedAtoms = s->padded_atoms;
    a.dt = s->dt; a.vscale = s->vscale; a.fscale = s->fscale; a.noisescale = s->noisescale;
    a.seed = s->seed; a.step = s->step;
    a.pos = (double4*) s->pos; a.vel = (double4*) s->vel; a.xp = (double4*) s->xp; a.oldx = (double4*) s->oldx;
    a.force = s->force; a.slotOfAtom = s->slot_of_atom; a.freeze = s->freeze_state;
    return a;
}

}  // namespace

static inline dim3 grid_for(int n) { return dim3((n + 127) / 128); }
#define BLOCK128 dim3(128)

extern "C" int ommhip_integrate_stage(int stage, const ommhip_integrator_state* s, void* stream) {
    IntArgs a = make_int_args(s);
    hipStream_t st = (hipStream_t) stream;
    if (a.numAtoms <= 0) return 0;
    switch (stage) {
        case OMMHIP_STAGE_VERLET_1: hipLaunchKernelGGL(k_verlet_part1, grid_for(a.numAtoms), BLOCK128, 0, st, a); break;
        case OMMHIP_STAGE_FINISH_POSITIONS: hipLaunchKernelGGL(k_finish_positions, grid_for(a.numAtoms), BLOCK128, 0, st, a); break;
        case OMMHIP_STAGE_LANGEVIN_1: hipLaunchKernelGGL(k_langevin_part1, grid_for(a.numAtoms), BLOCK128, 0, st, a); break;
        case OMMHIP_STAGE_LMIDDLE_1: hipLaunchKernelGGL(k_lmiddle_part1, grid_for(a.numAtoms), BLOCK128, 0, st, a); break;
        case OMMHIP_STAGE_LMIDDLE_2: hipLaunchKernelGGL(k_lmiddle_part2, grid_for(a.numAtoms), BLOCK128, 0, st, a); break;
        case OMMHIP_STAGE_LMIDDLE_3: hipLaunchKernelGGL(k_lmiddle_part3, grid_for(a.numAtoms), BLOCK128, 0, st, a); break;
        default: return 1;
    }
    return (int) hipGetLastError();
}

extern "C" int ommhip_integrate_fused(int integrator, const ommhip_integrator_state* s, const ommhip_step_units* units, void* stream) {
    IntArgs a = make_int_args(s);
    if (units->num_units <= 0) return 0;
    UnitArgs u;
    u.numUnits = units->num_units; u.maxIterations = units->max_iterations; u.removeCm = units->remove_cm;
    u.tol = units->tol; u.invTotalMass = units->inv_total_mass;
    u.atoms = (const int4*) units->atoms; u.dist = (const double4*) units->dist; u.cm = units->cm_scratch;
    u.posWire = (uint4*) units->pos_wire;
    for (int k = 0; k < 3; k++) { u.invBox[k] = units->box_len[k] > 0 ? 1.0 / units->box_len[k] : 0.0; u.skew[k] = units->box_skew[k]; }
    u.ranks = units->ranks; u.rank = units->rank; u.slotsPerRank = units->slots_per_rank; u.trailerSlot = units->trailer_slot;
    u.ddFlags = units->dd_flags;
    hipStream_t st = (hipStream_t) stream;
    const dim3 grid = grid_for(u.numUnits);
    switch (integrator) {
        case OMMHIP_INTEGRATOR_VERLET:
            if (units->small_units) hipLaunchKernelGGL((k_step_units<0, true>), grid, BLOCK128, 0, st, a, u); else hipLaunchKernelGGL((k_step_units<0, false>), grid, BLOCK128, 0, st, a, u);
            break;
        case OMMHIP_INTEGRATOR_LANGEVIN:
            if (units->small_units) hipLaunchKernelGGL((k_step_units<1, true>), grid, BLOCK128, 0, st, a, u); else hipLaunchKernelGGL((k_step_units<1, false>), grid, BLOCK128, 0, st, a, u);
            break;
        case OMMHIP_INTEGRATOR_LANGEVIN_MIDDLE:
            if (units->small_units) hipLaunchKernelGGL((k_step_units<2, true>), grid, BLOCK128, 0, st, a, u); else hipLaunchKernelGGL((k_step_units<2, false>), grid, BLOCK128, 0, st, a, u);
            break;
        default: return 1;
    }
    return (int) hipGetLastError();
}

extern "C" int ommhip_shifted_velocities(const ommhip_integrator_state* s, double shift, void* out_d, void* stream) {
    IntArgs a = make_int_args(s);
    if (a.numAtoms <= 0) return 0;
    hipLaunchKernelGGL(k_shifted_velocities, grid_for(a.numAtoms), BLOCK128, 0, (hipStream_t) stream, a, shift, (double4*) out_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_kinetic_energy(const void* vel_d, const int* atom_of_slot_d, int first, int end, double* scratch_d, double* result_d, void* stream) {
    int blocks = (end - first + 1023) / 1024;
    blocks = blocks < 1 ? 1 : (blocks > KE_PARTIALS ? KE_PARTIALS : blocks);
    hipLaunchKernelGGL(k_kinetic_energy_partial, dim3(blocks), dim3(256), 0, (hipStream_t) stream, (const double4*) vel_d, atom_of_slot_d, first, end, scratch_d);
    hipLaunchKernelGGL(k_kinetic_energy_final, dim3(1), dim3(256), 0, (hipStream_t) stream, (const double*) scratch_d, blocks, result_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_settle(int num_clusters, const int* atoms_d, const double* dist_d, const void* pos_d, void* target_d,
                             const void* vel_mass_d, int velocities, void* stream) {
    return ommhip_constrain_clusters(0, nullptr, nullptr, num_clusters, atoms_d, dist_d, pos_d, target_d, vel_mass_d, velocities, 0.0, 0, stream);
}

extern "C" int ommhip_shake(int num_clusters, const int* atoms_d, const double* dist_d, const void* pos_d, void* target_d,
                            const void* vel_mass_d, int velocities, double tol, int max_iterations, void* stream) {
    return ommhip_constrain_clusters(num_clusters, atoms_d, dist_d, 0, nullptr, nullptr, pos_d, target_d, vel_mass_d, velocities, tol, max_iterations, stream);
}

extern "C" int ommhip_constrain_clusters(int num_shake, const int* shake_atoms_d, const double* shake_dist_d,
                                         int num_settle, const int* settle_atoms_d, const double* settle_dist_d,
                                         const void* pos_d, void* target_d, const void* vel_mass_d, int velocities,
                                         double tol, int max_iterations, void* stream) {
    if (num_shake <= 0 && num_settle <= 0) return 0;
    ShakeArgs sh;
    sh.numClusters = num_shake > 0 ? num_shake : 0; sh.maxIterations = max_iterations; sh.tol = tol;
    sh.atoms = (const int4*) shake_atoms_d; sh.dist = (const double4*) shake_dist_d; sh.pos = (const double4*) pos_d;
    sh.target = (double4*) target_d; sh.velMass = (const double4*) vel_mass_d;
    SettleArgs se;
    se.numClusters = num_settle > 0 ? num_settle : 0; se.atoms = (const int4*) settle_atoms_d; se.dist = (const double2*) settle_dist_d;
    se.pos = (const double4*) pos_d; se.xp = (double4*) target_d; se.vel = (double4*) target_d; se.velMass = (const double4*) vel_mass_d;
    const int shakeBlocks = (sh.numClusters + 127) / 128, settleBlocks = (se.numClusters + 127) / 128;
    if (velocities) hipLaunchKernelGGL(k_constrain_clusters<true>, dim3(shakeBlocks + settleBlocks), BLOCK128, 0, (hipStream_t) stream, sh, se, shakeBlocks);
    else hipLaunchKernelGGL(k_constrain_clusters<false>, dim3(shakeBlocks + settleBlocks), BLOCK128, 0, (hipStream_t) stream, sh, se, shakeBlocks);
    return (int) hipGetLastError();
}

extern "C" int ommhip_ccma_iteration(const ommhip_ccma* c, const void* pos_d, void* target_d, const void* vel_mass_d,
                                     int velocities, double tol, int phase, void* stream) {
    if (c->num_constraints <= 0) return 0;
    CcmaArgs a;
    a.numConstraints = c->num_constraints; a.velocities = velocities; a.tol = tol;
    a.atoms = (const int2*) c->atoms; a.dist = c->distance; a.pos = (const double4*) pos_d; a.target = (double4*) target_d;
    a.velMass = (const double4*) vel_mass_d; a.delta = c->delta; a.delta2 = c->delta2;
    a.rowStart = c->row_start; a.col = c->col; a.value = c->value; a.converged = c->converged; a.resident = 0;
    hipStream_t st = (hipStream_t) stream;
    if (phase == 0) {
        hipMemsetAsync(c->converged, 0, sizeof(int), st);
        hipLaunchKernelGGL(k_ccma_delta, grid_for(a.numConstraints), BLOCK128, 0, st, a);
    }
    else {
        hipLaunchKernelGGL(k_ccma_multiply, grid_for(a.numConstraints), BLOCK128, 0, st, a);
        hipLaunchKernelGGL(k_ccma_update, grid_for(a.numConstraints), BLOCK128, 0, st, a);
    }
    return (int) hipGetLastError();
}

// Device-resident CCMA: `iterations` iterations enqueued back to back; each one leaves at once when an earlier one found every
// constraint converged (converged[2], set by the last workgroup of its delta kernel).  The caller zeroes converged[0..3] before
// the first batch and reads converged[2] after each batch -- the loop of platforms/cuda/src/CudaIntegrationUtilities.cpp:94-130,
// which looks at a host-mapped flag every few iterations instead of after every one.
extern "C" int ommhip_ccma_iterations(const ommhip_ccma* c, const void* pos_d, void* target_d, const void* vel_mass_d,
                                      int velocities, double tol, int iterations, void* stream) {
    if (c->num_constraints <= 0) return 0;
    CcmaArgs a;
    a.numConstraints = c->num_constraints; a.velocities = velocities; a.tol = tol;
    a.atoms = (const int2*) c->atoms; a.dist = c->distance; a.pos = (const double4*) pos_d; a.target = (double4*) target_d;
    a.velMass = (const double4*) vel_mass_d; a.delta = c->delta; a.delta2 = c->delta2;
    a.rowStart = c->row_start; a.col = c->col; a.value = c->value; a.converged = c->converged; a.resident = 1;
    hipStream_t st = (hipStream_t) stream;
    for (int i = 0; i < iterations; i++) {
        hipLaunchKernelGGL(k_ccma_delta, grid_for(a.numConstraints), BLOCK128, 0, st, a);
        hipLaunchKernelGGL(k_ccma_multiply, grid_for(a.numConstraints), BLOCK128, 0, st, a);
        hipLaunchKernelGGL(k_ccma_update, grid_for(a.numConstraints), BLOCK128, 0, st, a);
    }
    return (int) hipGetLastError();
}
