// The valence terms of the AMOEBA force field, one thread per term, double precision on the unwrapped double positions (atom order);
// results go to the fixed-point force buffer (slot order) like the terms of bonded.hip.
//
// The reference's Python layer writes these terms as Custom*Forces whose energy expressions its platforms differentiate symbolically and
// compile at run time (wrappers/python/openmm/app/forcefield.py:3368 bond, :3502 angle, :3565 in-plane angle, :3730 out-of-plane bend,
// :4039 pi-torsion, :4428 stretch-bend; the torsion-torsion map is a Force of the AMOEBA plugin:
// plugins/amoeba/platforms/reference/src/SimTKReference/AmoebaReferenceTorsionTorsionForce.cpp:283-530).  Here each expression is written
// once as a function of the atoms' coordinates over a forward-mode dual number (value + gradient with respect to the coordinates RELATIVE
// to one atom of the term: 6 to 15 partial derivatives), so the forces are the derivatives of exactly the energy that is evaluated and no
// gradient is derived by hand; the force on the reference atom is minus the sum of the others (translation invariance).  A System has a few
// thousand such terms: the launch is latency-bound, arithmetic is free, and double precision removes a source of parity noise.
// All lists of one call share ONE launch (every list owns a contiguous range of workgroups, as in bonded.hip).
#include "common.h"
#include "../../../include/openmm_hip_kernels.h"

using namespace omm;

namespace {

template <int N>
struct Dual {
    double v, d[N];
};
template <int N> __device__ __forceinline__ Dual<N> constant(double c) { Dual<N> r; r.v = c;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = 0; return r; }
template <int N> __device__ __forceinline__ Dual<N> variable(double value, int index) { Dual<N> r = constant<N>(value); r.d[index] = 1; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator+(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a) { Dual<N> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = -a.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, const Dual<N>& b) { Dual<N> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator*(const Dual<N>& a, double s) { Dual<N> r; r.v = a.v * s;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * s; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator-(const Dual<N>& a, double s) { Dual<N> r = a; r.v -= s; return r; }
// f(a) with the derivative f'(a) given: the chain rule
template <int N> __device__ __forceinline__ Dual<N> chain(const Dual<N>& a, double value, double slope) { Dual<N> r; r.v = value;
#pragma unroll
    for (int i = 0; i < N; i++) r.d[i] = a.d[i] * slope; return r; }
template <int N> __device__ __forceinline__ Dual<N> operator/(const Dual<N>& a, const Dual<N>& b) { const double inv = 1.0 / b.v; return a * chain(b, inv, -inv * inv); }
template <int N> __device__ __forceinline__ Dual<N> sqrt(const Dual<N>& a) { const double s = ::sqrt(a.v); return chain(a, s, s > 0 ? 0.5 / s : 0.0); }
// acos with the argument clamped to [-1, 1]; at the ends (a straight or folded angle) the derivative is taken a hair inside
template <int N> __device__ __forceinline__ Dual<N> acos(const Dual<N>& a) {
    const double c = fmin(1.0, fmax(-1.0, a.v));
    return chain(a, ::acos(c), -1.0 / ::sqrt(fmax(1.0 - c * c, 1e-24)));
}

template <int N> struct Vec { Dual<N> x, y, z; };
template <int N> __device__ __forceinline__ Vec<N> operator-(const Vec<N>& a, const Vec<N>& b) { Vec<N> r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
template <int N> __device__ __forceinline__ Vec<N> operator*(const Vec<N>& a, const Dual<N>& s) { Vec<N> r = {a.x * s, a.y * s, a.z * s}; return r; }
template <int N> __device__ __forceinline__ Dual<N> dot(const Vec<N>& a, const Vec<N>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <int N> __device__ __forceinline__ Vec<N> cross(const Vec<N>& a, const Vec<N>& b) { Vec<N> r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; }
template <int N> __device__ __forceinline__ Dual<N> norm(const Vec<N>& a) { return sqrt(dot(a, a)); }
// the angle between two vectors, radians (Lepton's angle() / pointangle() with the vertex subtracted)
template <int N> __device__ __forceinline__ Dual<N> angle_between(const Vec<N>& a, const Vec<N>& b) { return acos(dot(a, b) / sqrt(dot(a, a) * dot(b, b))); }

struct ValenceList {
    int kind, numTerms, firstBlock;
    const int* atoms;
    const double* params;
    const double* grids;
    double c[6];
};

struct ValenceArgs {
    int numLists, paddedAtoms, includeEnergy, energySlots;
    const double4* pos;
    const int* slotOfAtom;
    omm_fixed* force;
    double* energyBuffer;
    ValenceList list[OMMHIP_MAX_VALENCE_LISTS];
};

struct Ctx {
    const ValenceArgs& a;
    __device__ Ctx(const ValenceArgs& a_) : a(a_) {}
    __device__ __forceinline__ double3 at(int atom) const { const double4 p = a.pos[atom]; return make_double3(p.x, p.y, p.z); }
    __device__ __forceinline__ void add(int atom, double fx, double fy, double fz) const { add_force(a.force, a.paddedAtoms, a.slotOfAtom[atom], fx, fy, fz); }
    // atom `which` of the term as variables 3 * which ... of the dual numbers, relative to `origin`
    template <int N> __device__ __forceinline__ Vec<N> rel(int atom, double3 origin, int which) const {
        const double3 p = at(atom);
        Vec<N> r = {variable<N>(p.x - origin.x, 3 * which), variable<N>(p.y - origin.y, 3 * which + 1), variable<N>(p.z - origin.z, 3 * which + 2)};
        return r;
    }
    // forces from the gradient: atoms[0 .. M) carry the variables, `origin` takes minus their sum
    template <int N> __device__ __forceinline__ void scatter(const Dual<N>& e, const int* atoms, int originAtom) const {
        double sx = 0, sy = 0, sz = 0;
#pragma unroll
        for (int k = 0; k < N / 3; k++) {
            add(atoms[k], -e.d[3 * k], -e.d[3 * k + 1], -e.d[3 * k + 2]);
            sx += e.d[3 * k]; sy += e.d[3 * k + 1]; sz += e.d[3 * k + 2];
        }
        add(originAtom, sx, sy, sz);
    }
};

__device__ __forceinline__ double3 sub3(double3 a, double3 b) { return make_double3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ double dot3(double3 a, double3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double3 cross3(double3 a, double3 b) { return make_double3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// k (d^2 + c0 d^3 + c1 d^4), d = r - r0
__device__ __forceinline__ double term_poly_bond(const Ctx& c, const ValenceList& l, int t) {
    const int i = l.atoms[2 * t], j = l.atoms[2 * t + 1];
    const double r0 = l.params[2 * t], k = l.params[2 * t + 1];
    const double3 d = sub3(c.at(j), c.at(i));
    const double r = ::sqrt(dot3(d, d)), x = r - r0;
    const double dEdR = r > 0.0 ? k * x * (2.0 + x * (3.0 * l.c[0] + x * 4.0 * l.c[1])) / r : 0.0;
    c.add(i, dEdR * d.x, dEdR * d.y, dEdR * d.z);
    c.add(j, -dEdR * d.x, -dEdR * d.y, -dEdR * d.z);
    return k * x * x * (1.0 + x * (l.c[0] + x * l.c[1]));
}

// the polynomial of the three angle-like terms: value and derivative of x^2 + c0 x^3 + c1 x^4 + c2 x^5 + c3 x^6
__device__ __forceinline__ void sextic(const double* c, double x, double& value, double& slope) {
    value = x * x * (1.0 + x * (c[0] + x * (c[1] + x * (c[2] + x * c[3]))));
    slope = x * (2.0 + x * (3.0 * c[0] + x * (4.0 * c[1] + x * (5.0 * c[2] + x * 6.0 * c[3]))));
}
template <int N> __device__ __forceinline__ Dual<N> sextic_energy(const double* c, const Dual<N>& x, double k) {
    double value, slope;
    sextic(c, x.v, value, slope);
    return chain(x, k * value, k * slope);
}

// k sextic(d), d = c4 theta - theta0 (c4 = 180 / pi: the parameters are in degrees)
__device__ __forceinline__ double term_poly_angle(const Ctx& c, const ValenceList& l, int t) {
    const int atoms[3] = {l.atoms[3 * t], l.atoms[3 * t + 2], l.atoms[3 * t + 1]};      // the centre last: it is the origin
    const double3 o = c.at(atoms[2]);
    const Vec<6> u = c.rel<6>(atoms[0], o, 0), w = c.rel<6>(atoms[1], o, 1);
    const Dual<6> e = sextic_energy(l.c, angle_between(u, w) * l.c[4] - l.params[2 * t], l.params[2 * t + 1]);
    c.scatter(e, atoms, atoms[2]);
    return e.v;
}

// The foot of the perpendicular from b onto the plane through a, c and the origin (the fourth atom): b - n (n . (b - c)), n the unit normal
template <int N> __device__ __forceinline__ Vec<N> project_on_plane(const Vec<N>& a, const Vec<N>& b, const Vec<N>& c) {
    const Vec<N> p = cross(a, c);
    const Dual<N> inv = constant<N>(1.0) / norm(p);
    const Vec<N> n = p * inv;
    return b - n * dot(n, b - c);
}

// in-plane angle at a trigonal centre (atom 2 of 1-2-3, fourth atom 4): the angle 1 - P - 3 with P the centre projected onto the plane 1-3-4
__device__ __forceinline__ double term_inplane_angle(const Ctx& c, const ValenceList& l, int t) {
    const int atoms[4] = {l.atoms[4 * t], l.atoms[4 * t + 1], l.atoms[4 * t + 2], l.atoms[4 * t + 3]};
    const double3 o = c.at(atoms[3]);
    const Vec<9> a = c.rel<9>(atoms[0], o, 0), b = c.rel<9>(atoms[1], o, 1), cc = c.rel<9>(atoms[2], o, 2);
    const Vec<9> p = project_on_plane(a, b, cc);
    const Dual<9> e = sextic_energy(l.c, angle_between(a - p, cc - p) * l.c[4] - l.params[2 * t], l.params[2 * t + 1]);
    c.scatter(e, atoms, atoms[3]);
    return e.v;
}

// Allinger out-of-plane bend: the angle at atom 4 between the centre (atom 2) and its projection onto the plane 1-3-4, in degrees
__device__ __forceinline__ double term_out_of_plane_bend(const Ctx& c, const ValenceList& l, int t) {
    const int atoms[4] = {l.atoms[4 * t], l.atoms[4 * t + 1], l.atoms[4 * t + 2], l.atoms[4 * t + 3]};
    const double3 o = c.at(atoms[3]);
    const Vec<9> a = c.rel<9>(atoms[0], o, 0), b = c.rel<9>(atoms[1], o, 1), cc = c.rel<9>(atoms[2], o, 2);
    const Vec<9> p = project_on_plane(a, b, cc);
    const Dual<9> e = sextic_energy(l.c, angle_between(b, p) * l.c[4], l.params[t]);
    c.scatter(e, atoms, atoms[3]);
    return e.v;
}

// (k1 (|12| - r12) + k2 (|23| - r23)) c0 (angle(1, 2, 3) - theta0); parameters r12, r23, theta0 [rad], k1, k2
__device__ __forceinline__ double term_stretch_bend(const Ctx& c, const ValenceList& l, int t) {
    const int atoms[3] = {l.atoms[3 * t], l.atoms[3 * t + 2], l.atoms[3 * t + 1]};
    const double* par = l.params + 5 * t;
    const double3 o = c.at(atoms[2]);
    const Vec<6> u = c.rel<6>(atoms[0], o, 0), w = c.rel<6>(atoms[1], o, 1);
    const Dual<6> e = ((norm(u) - par[0]) * par[3] + (norm(w) - par[1]) * par[4]) * ((angle_between(u, w) - par[2]) * l.c[0]);
    c.scatter(e, atoms, atoms[2]);
    return e.v;
}

// 2 k sin^2(phi), phi the angle between the normals of the planes 1-2-4 (at atom 4... the substituents of atom 3) and 5-6-3, seen along
// the bond 3-4: pointdihedral(3 + c1, 3, 4, 4 + c2) with c1 = (1 - 4) x (2 - 4), c2 = (5 - 3) x (6 - 3).  sin^2 = 1 - cos^2: no sign needed.
__device__ __forceinline__ double term_pi_torsion(const Ctx& c, const ValenceList& l, int t) {
    const int atoms[6] = {l.atoms[6 * t], l.atoms[6 * t + 1], l.atoms[6 * t + 3], l.atoms[6 * t + 4], l.atoms[6 * t + 5], l.atoms[6 * t + 2]};   // atom 3 last: the origin
    const double3 o = c.at(atoms[5]);
    const Vec<15> p1 = c.rel<15>(atoms[0], o, 0), p2 = c.rel<15>(atoms[1], o, 1), p4 = c.rel<15>(atoms[2], o, 2), p5 = c.rel<15>(atoms[3], o, 3), p6 = c.rel<15>(atoms[4], o, 4);
    const Vec<15> c1 = cross(p1 - p4, p2 - p4), c2 = cross(p5, p6);
    // dihedral of the points (c1, 0, p4, p4 + c2): v0 = c1, v1 = p4, v2 = -c2; the normals of its two planes
    const Vec<15> n0 = cross(c1, p4), n1 = cross(p4, c2);
    const Dual<15> cosPhi = dot(n0, n1) / sqrt(dot(n0, n0) * dot(n1, n1));
    const Dual<15> e = (constant<15>(1.0) - cosPhi * cosPhi) * (2.0 * l.params[t]);
    c.scatter(e, atoms, atoms[5]);
    return e.v;
}

// A dihedral angle a-b-c-d in (-pi, pi] with the sign convention of the reference's torsion-torsion code (AmoebaReferenceTorsionTorsionForce.cpp:
// 350-380: positive when (b - a) . ((c - b) x (d - c)) >= 0), and the forces -dE/dphi grad(phi) on its four atoms added for a given dE/dphi.
struct Dihedral {
    double3 ba, cb, dc, t, u;
    double rt2, ru2, rcb, phi;
    __device__ __forceinline__ Dihedral(double3 a, double3 b, double3 c, double3 d) {
        ba = sub3(b, a); cb = sub3(c, b); dc = sub3(d, c);
        t = cross3(ba, cb); u = cross3(cb, dc);
        rt2 = dot3(t, t); ru2 = dot3(u, u); rcb = ::sqrt(dot3(cb, cb));
        const double cosine = dot3(t, u) / ::sqrt(rt2 * ru2);
        phi = ::acos(fmin(1.0, fmax(-1.0, cosine)));
        if (dot3(ba, u) < 0.0) phi = -phi;
    }
    // d(phi)/d(atom): the classic result -- the end atoms move phi along the normals of their planes, the inner atoms take what keeps the sum
    // and the torque zero
    __device__ __forceinline__ void gradient(double3& ga, double3& gb, double3& gc, double3& gd) const {
        const double fa = -rcb / rt2, fd = rcb / ru2;
        ga = make_double3(fa * t.x, fa * t.y, fa * t.z);
        gd = make_double3(fd * u.x, fd * u.y, fd * u.z);
        const double pa = dot3(ba, cb) / (rcb * rcb), pd = dot3(dc, cb) / (rcb * rcb);
        const double3 s = make_double3(pd * gd.x - pa * ga.x, pd * gd.y - pa * ga.y, pd * gd.z - pa * ga.z);
        gb = make_double3(s.x - ga.x, s.y - ga.y, s.z - ga.z);
        gc = make_double3(-s.x - gd.x, -s.y - gd.y, -s.z - gd.z);
    }
};

// The bicubic Hermite patch over one cell of the (phi, psi) map: the unique bicubic with the tabulated f, df/dx, df/dy, d2f/dxdy at the four
// corners (Numerical Recipes' bcuint, written through the cubic Hermite basis).  -> f, df/dx, df/dy at (x, y) [degrees].
__device__ __forceinline__ void bicubic(const double* grid, int n, double x, double y, double& f, double& fx, double& fy) {
    const double x0 = grid[0], y0 = grid[1];
    const double perDegree = (n - 1) / 360.0;
    int ix = (int) ((x - x0) * perDegree + 1e-6), iy = (int) ((y - y0) * perDegree + 1e-6);
    ix = ix < 0 ? 0 : (ix > n - 2 ? n - 2 : ix); iy = iy < 0 ? 0 : (iy > n - 2 ? n - 2 : iy);
    const double* c00 = grid + 6 * ((size_t) ix * n + iy);
    const double* c10 = c00 + 6 * n;
    const double* c01 = c00 + 6;
    const double* c11 = c10 + 6;
    const double wx = c10[0] - c00[0], wy = c01[1] - c00[1];
    const double s = (x - c00[0]) / wx, r = (y - c00[1]) / wy;
    // Hermite basis on [0, 1]: value at 0, value at 1, slope at 0, slope at 1 -- and their derivatives
    const double hs[4] = {(2 * s - 3) * s * s + 1, (3 - 2 * s) * s * s, ((s - 2) * s + 1) * s, (s - 1) * s * s};
    const double ds[4] = {6 * s * (s - 1), 6 * s * (1 - s), (3 * s - 4) * s + 1, (3 * s - 2) * s};
    const double hr[4] = {(2 * r - 3) * r * r + 1, (3 - 2 * r) * r * r, ((r - 2) * r + 1) * r, (r - 1) * r * r};
    const double dr[4] = {6 * r * (r - 1), 6 * r * (1 - r), (3 * r - 4) * r + 1, (3 * r - 2) * r};
    const double* corner[2][2] = {{c00, c01}, {c10, c11}};
    f = fx = fy = 0;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const double* g = corner[a][b];
            const double v = g[2], vx = g[3] * wx, vy = g[4] * wy, vxy = g[5] * wx * wy;
            f += v * hs[a] * hr[b] + vx * hs[2 + a] * hr[b] + vy * hs[a] * hr[2 + b] + vxy * hs[2 + a] * hr[2 + b];
            fx += v * ds[a] * hr[b] + vx * ds[2 + a] * hr[b] + vy * ds[a] * hr[2 + b] + vxy * ds[2 + a] * hr[2 + b];
            fy += v * hs[a] * dr[b] + vx * hs[2 + a] * dr[b] + vy * hs[a] * dr[2 + b] + vxy * hs[2 + a] * dr[2 + b];
        }
    fx /= wx; fy /= wy;
}

// torsion-torsion: atoms a-b-c-d-e and the chirality marker (or -1); params: where the term's map starts in `grids` (in doubles) and its points per axis.  E = map(phi(a,b,c,d), psi(b,c,d,e)), both
// angles negated when the centre's substituents (marker, b, d) are left-handed (AmoebaReferenceTorsionTorsionForce.cpp:232-262, 398-410).
__device__ __forceinline__ double term_torsion_torsion(const Ctx& c, const ValenceList& l, int t) {
    const int* at = l.atoms + 6 * t;
    const double3 a = c.at(at[0]), b = c.at(at[1]), cc = c.at(at[2]), d = c.at(at[3]), e = c.at(at[4]);
    const Dihedral phi(a, b, cc, d), psi(b, cc, d, e);
    if (phi.rt2 * phi.ru2 <= 0.0 || psi.rt2 * psi.ru2 <= 0.0) return 0.0;
    double sign = 1.0;
    if (at[5] >= 0) {
        const double3 ca = sub3(c.at(at[5]), cc), cb = sub3(b, cc), cd = sub3(d, cc);
        if (dot3(ca, cross3(cb, cd)) < 0.0) sign = -1.0;
    }
    const double toDegrees = 57.29577951308232;
    double energy, dE1, dE2;
    bicubic(l.grids + (size_t) l.params[2 * t], (int) l.params[2 * t + 1], sign * toDegrees * phi.phi, sign * toDegrees * psi.phi, energy, dE1, dE2);
    dE1 *= sign * toDegrees; dE2 *= sign * toDegrees;
    double3 g1[4], g2[4];
    phi.gradient(g1[0], g1[1], g1[2], g1[3]);
    psi.gradient(g2[0], g2[1], g2[2], g2[3]);
    c.add(at[0], -dE1 * g1[0].x, -dE1 * g1[0].y, -dE1 * g1[0].z);
    c.add(at[1], -dE1 * g1[1].x - dE2 * g2[0].x, -dE1 * g1[1].y - dE2 * g2[0].y, -dE1 * g1[1].z - dE2 * g2[0].z);
    c.add(at[2], -dE1 * g1[2].x - dE2 * g2[1].x, -dE1 * g1[2].y - dE2 * g2[1].y, -dE1 * g1[2].z - dE2 * g2[1].z);
    c.add(at[3], -dE1 * g1[3].x - dE2 * g2[2].x, -dE1 * g1[3].y - dE2 * g2[2].y, -dE1 * g1[3].z - dE2 * g2[2].z);
    c.add(at[4], -dE2 * g2[3].x, -dE2 * g2[3].y, -dE2 * g2[3].z);
    return energy;
}

__global__ __launch_bounds__(128) void k_valence(ValenceArgs a) {
    __shared__ double partial[2];
    const int block = blockIdx.x;
    int li = 0;
#pragma unroll
    for (int i = 1; i < OMMHIP_MAX_VALENCE_LISTS; i++)
        if (i < a.numLists && block >= a.list[i].firstBlock) li = i;
    const ValenceList& l = a.list[li];
    const int t = (block - l.firstBlock) * 128 + threadIdx.x;
    double energy = 0;
    if (t < l.numTerms) {
        const Ctx c(a);
        switch (l.kind) {
            case OMMHIP_VALENCE_POLY_BOND: energy = term_poly_bond(c, l, t); break;
            case OMMHIP_VALENCE_POLY_ANGLE: energy = term_poly_angle(c, l, t); break;
            case OMMHIP_VALENCE_INPLANE_ANGLE: energy = term_inplane_angle(c, l, t); break;
            case OMMHIP_VALENCE_OUT_OF_PLANE_BEND: energy = term_out_of_plane_bend(c, l, t); break;
            case OMMHIP_VALENCE_STRETCH_BEND: energy = term_stretch_bend(c, l, t); break;
            case OMMHIP_VALENCE_PI_TORSION: energy = term_pi_torsion(c, l, t); break;
            default: energy = term_torsion_torsion(c, l, t); break;
        }
    }
    if (a.includeEnergy) {
        energy = wave_sum(energy);
        if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = energy;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&a.energyBuffer[block % a.energySlots], partial[0] + partial[1]);
    }
}

}  // namespace

extern "C" int ommhip_valence_forces(int num_lists, const ommhip_valence_list* lists, const void* pos_d, const int* slot_of_atom_d, int padded_atoms,
                                     long long* force_d, double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    if (num_lists > OMMHIP_MAX_VALENCE_LISTS) return 1;
    ValenceArgs a;
    a.numLists = 0; a.paddedAtoms = padded_atoms; a.includeEnergy = include_energy; a.energySlots = energy_slots;
    a.pos = (const double4*) pos_d; a.slotOfAtom = slot_of_atom_d; a.force = force_d; a.energyBuffer = energy_buffer_d;
    int blocks = 0;
    for (int i = 0; i < num_lists; i++) {
        if (lists[i].num_terms <= 0) continue;
        if (lists[i].kind < OMMHIP_VALENCE_POLY_BOND || lists[i].kind > OMMHIP_VALENCE_TORSION_TORSION) return 1;
        if (lists[i].kind == OMMHIP_VALENCE_TORSION_TORSION && lists[i].grids == NULL) return 1;
        ValenceList& l = a.list[a.numLists++];
        l.kind = lists[i].kind; l.numTerms = lists[i].num_terms; l.firstBlock = blocks; l.atoms = lists[i].atoms; l.params = lists[i].params;
        l.grids = lists[i].grids;
        for (int k = 0; k < 6; k++) l.c[k] = lists[i].coefficients[k];
        blocks += (l.numTerms + 127) / 128;
    }
    if (blocks == 0) return 0;
    for (int i = a.numLists; i < OMMHIP_MAX_VALENCE_LISTS; i++) { a.list[i] = a.list[0]; a.list[i].numTerms = 0; a.list[i].firstBlock = blocks; }
    hipLaunchKernelGGL(k_valence, dim3(blocks), dim3(128), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}
