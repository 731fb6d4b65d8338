// State conversion kernels: double-precision atom-ordered state <-> float slot-ordered working set,
// fixed-point force import/export, energy reduction.  See include/openmm_hip_kernels.h.
#include "common.h"
#include "../../../include/openmm_hip_kernels.h"

using namespace omm;

namespace {

__global__ void k_positions_to_posq(const double4* __restrict__ pos, const int4* __restrict__ wrap, const int* __restrict__ atomOfSlot,
                                    int paddedAtoms, BoxD box, float4* __restrict__ posq) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= paddedAtoms) return;
    int a = atomOfSlot[s];
    float4 out = posq[s];
    if (a >= 0) {
        double4 p = pos[a];
        int4 w = wrap[a];
        // subtract the periodic image chosen at the last reorder so that the float copy stays near the primary cell
        double x = p.x - (w.x * box.ax + w.y * box.bx + w.z * box.cx);
        double y = p.y - (w.y * box.by + w.z * box.cy);
        double z = p.z - (w.z * box.cz);
        out.x = (float) x; out.y = (float) y; out.z = (float) z;
    }
    else {
        out.x = 0.f; out.y = 0.f; out.z = 0.f; out.w = 0.f;
    }
    posq[s] = out;
}

__global__ void k_set_slot_params(const double* __restrict__ charge, const double* __restrict__ sigma, const double* __restrict__ epsilon,
                                  const int* __restrict__ atomOfSlot, int paddedAtoms, float4* __restrict__ posq, float2* __restrict__ sigEps) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= paddedAtoms) return;
    int a = atomOfSlot[s];
    float q = 0.f;
    float2 se = make_float2(0.f, 0.f);
    if (a >= 0) {
        q = (float) charge[a];
        se = make_float2((float) (0.5 * sigma[a]), (float) (2.0 * sqrt(epsilon[a])));
    }
    posq[s].w = q;
    sigEps[s] = se;
}

__global__ void k_forces_to_double(const omm_fixed* __restrict__ force, const int* __restrict__ slotOfAtom, int numAtoms, int paddedAtoms,
                                   double* __restrict__ out) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= numAtoms) return;
    int s = slotOfAtom[a];
    out[3 * a] = from_fixed(force[s]);
    out[3 * a + 1] = from_fixed(force[s + paddedAtoms]);
    out[3 * a + 2] = from_fixed(force[s + 2 * paddedAtoms]);
}

__global__ void k_add_forces_from_double(const double* __restrict__ in, const int* __restrict__ slotOfAtom, int numAtoms, int paddedAtoms,
                                         omm_fixed* __restrict__ force) {
    int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= numAtoms) return;
    int s = slotOfAtom[a];
    force[s] += to_fixed(in[3 * a]);
    force[s + paddedAtoms] += to_fixed(in[3 * a + 1]);
    force[s + 2 * paddedAtoms] += to_fixed(in[3 * a + 2]);
}

__global__ __launch_bounds__(256) void k_reduce_energy(double* __restrict__ buffer, int n, double* __restrict__ result) {
    __shared__ double partial[4];
    double sum = 0;
    for (int i = threadIdx.x; i < n; i += 256) { sum += buffer[i]; buffer[i] = 0; }
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) result[0] = partial[0] + partial[1] + partial[2] + partial[3];
}

// Zero two buffers (16-byte granules) in one launch: the force accumulator and the PME charge grid.
__global__ __launch_bounds__(256) void k_clear2(uint4* __restrict__ a, size_t na, uint4* __restrict__ b, size_t nb) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < na + nb; i += stride) {
        if (i < na) a[i] = z;
        else b[i - na] = z;
    }
}

__global__ void k_posq_with_weights(const float4* __restrict__ posq, const double* __restrict__ weight, const int* __restrict__ atomOfSlot, int paddedAtoms, float4* __restrict__ dst) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= paddedAtoms) return;
    const float4 p = posq[s];
    const int a = atomOfSlot[s];
    dst[s] = make_float4(p.x, p.y, p.z, a >= 0 ? (float) weight[a] : 0.f);
}

// positions -> wire records (fixed-point fractions of the box edges), slot order
__global__ void k_encode_wire(const double4* __restrict__ pos, const int* __restrict__ atomOfSlot, int slot0, int slot1, double ix, double iy, double iz, double bx, double cx, double cy, uint4* __restrict__ wire) {
    const int s = slot0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slot1) return;
    const int a = atomOfSlot[s];
    if (a < 0) return;
    const double4 p = pos[a];
    // coefficients of the box vectors c, b, a in turn (a rectangular box: bx = cx = cy = 0, the same numbers as x / ax ...)
    const double fz0 = p.z * iz, fy0 = (p.y - fz0 * cy) * iy;
    double fx = (p.x - fy0 * bx - fz0 * cx) * ix, fy = fy0, fz = fz0;
    fx -= floor(fx); fy -= floor(fy); fz -= floor(fz);
    wire[s] = make_uint4((unsigned) (unsigned long long) (fx * 4294967296.0), (unsigned) (unsigned long long) (fy * 4294967296.0),
                         (unsigned) (unsigned long long) (fz * 4294967296.0), 0u);
}

// the fourth double of every rank's trailer record (the "an atom of mine is near the drift margin" flag) back to zero: at a re-sort
__global__ void k_clear_trailer_flags(uint4* __restrict__ wire, int ranks, int slotsPerRank, int trailerSlot) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < ranks) ((double4*) (wire + (size_t) r * slotsPerRank + trailerSlot))->w = 0.0;
}

// slot-ordered <-> atom-ordered copies of a double4 array (all-gather buffers of the decomposed run)
__global__ void k_pack_slots(const double4* __restrict__ src, const int* __restrict__ atomOfSlot, int slot0, int slot1, double4* __restrict__ dst) {
    const int s = slot0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slot1) return;
    const int a = atomOfSlot[s];
    if (a >= 0) dst[s] = src[a];
}
__global__ void k_unpack_slots(const double4* __restrict__ src, const int* __restrict__ atomOfSlot, int slot0, int slot1, double4* __restrict__ dst) {
    const int s = slot0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= slot1) return;
    const int a = atomOfSlot[s];
    if (a >= 0) dst[a] = src[s];
}

// ReferenceMonteCarloBarostat.cpp:68-104, one thread per molecule
__global__ void k_scale_molecule_centers(int numMolecules, const int* __restrict__ molStart, const int* __restrict__ molAtoms, double4* __restrict__ pos,
                                         BoxD box, double sx, double sy, double sz) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= numMolecules) return;
    const int b = molStart[m], e = molStart[m + 1];
    double cx = 0, cy = 0, cz = 0;
    for (int i = b; i < e; i++) { const double4 p = pos[molAtoms[i]]; cx += p.x; cy += p.y; cz += p.z; }
    const double inv = 1.0 / (e - b);
    cx *= inv; cy *= inv; cz *= inv;
    double nx = cx, ny = cy, nz = cz;
    double f = floor(nz / box.cz); nx -= f * box.cx; ny -= f * box.cy; nz -= f * box.cz;
    f = floor(ny / box.by); nx -= f * box.bx; ny -= f * box.by;
    f = floor(nx / box.ax); nx -= f * box.ax;
    const double ox = nx * sx - cx, oy = ny * sy - cy, oz = nz * sz - cz;
    for (int i = b; i < e; i++) { double4 p = pos[molAtoms[i]]; p.x += ox; p.y += oy; p.z += oz; pos[molAtoms[i]] = p; }
}

BoxD make_boxd(const double* bv) {
    BoxD b; b.ax = bv[0]; b.bx = bv[1]; b.by = bv[2]; b.cx = bv[3]; b.cy = bv[4]; b.cz = bv[5];
    return b;
}

}  // namespace

extern "C" int ommhip_positions_to_posq(const void* pos_d, const void* wrap_d, const int* atom_of_slot_d, int padded_atoms,
                                        const double box[6], void* posq_d, void* stream) {
    hipLaunchKernelGGL(k_positions_to_posq, dim3((padded_atoms + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                       (const double4*) pos_d, (const int4*) wrap_d, atom_of_slot_d, padded_atoms, make_boxd(box), (float4*) posq_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_set_slot_params(const double* charge_d, const double* sigma_d, const double* epsilon_d, const int* atom_of_slot_d,
                                      int padded_atoms, void* posq_d, void* sig_eps_d, void* stream) {
    hipLaunchKernelGGL(k_set_slot_params, dim3((padded_atoms + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                       charge_d, sigma_d, epsilon_d, atom_of_slot_d, padded_atoms, (float4*) posq_d, (float2*) sig_eps_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_forces_to_double(const long long* force_d, const int* slot_of_atom_d, int num_atoms, int padded_atoms, double* out_d, void* stream) {
    hipLaunchKernelGGL(k_forces_to_double, dim3((num_atoms + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                       (const omm_fixed*) force_d, slot_of_atom_d, num_atoms, padded_atoms, out_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_add_forces_from_double(const double* in_d, const int* slot_of_atom_d, int num_atoms, int padded_atoms, long long* force_d, void* stream) {
    hipLaunchKernelGGL(k_add_forces_from_double, dim3((num_atoms + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                       in_d, slot_of_atom_d, num_atoms, padded_atoms, (omm_fixed*) force_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_clear2(void* a_d, size_t a_bytes, void* b_d, size_t b_bytes, void* stream) {
    // both sizes must be multiples of 16 bytes (all buffers of this library are)
    if ((a_bytes & 15) != 0 || (b_bytes & 15) != 0) return 1;
    const size_t n = a_bytes / 16 + b_bytes / 16;
    if (n == 0) return 0;
    int blocks = (int) ((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_clear2, dim3(blocks), dim3(256), 0, (hipStream_t) stream, (uint4*) a_d, a_bytes / 16, (uint4*) b_d, b_bytes / 16);
    return (int) hipGetLastError();
}

extern "C" int ommhip_reduce_energy(double* buffer_d, int n, double* result_d, void* stream) {
    hipLaunchKernelGGL(k_reduce_energy, dim3(1), dim3(256), 0, (hipStream_t) stream, buffer_d, n, result_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_pack_slots(const void* src_atom_order_d, const int* atom_of_slot_d, int slot0, int slot1, void* dst_slot_order_d, void* stream) {
    if (slot1 <= slot0) return 0;
    hipLaunchKernelGGL(k_pack_slots, dim3((slot1 - slot0 + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                       (const double4*) src_atom_order_d, atom_of_slot_d, slot0, slot1, (double4*) dst_slot_order_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_unpack_slots(const void* src_slot_order_d, const int* atom_of_slot_d, int slot0, int slot1, void* dst_atom_order_d, void* stream) {
    if (slot1 <= slot0) return 0;
    hipLaunchKernelGGL(k_unpack_slots, dim3((slot1 - slot0 + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                       (const double4*) src_slot_order_d, atom_of_slot_d, slot0, slot1, (double4*) dst_atom_order_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_scale_molecule_centers(int num_molecules, const int* mol_start_d, const int* mol_atoms_d, void* pos_d,
                                             const double box[6], double sx, double sy, double sz, void* stream) {
    if (num_molecules <= 0) return 0;
    hipLaunchKernelGGL(k_scale_molecule_centers, dim3((num_molecules + 127) / 128), dim3(128), 0, (hipStream_t) stream,
                       num_molecules, mol_start_d, mol_atoms_d, (double4*) pos_d, make_boxd(box), sx, sy, sz);
    return (int) hipGetLastError();
}

extern "C" int ommhip_posq_with_weights(const void* posq_d, const double* weight_d, const int* atom_of_slot_d, int padded_atoms, void* dst_d, void* stream) {
    hipLaunchKernelGGL(k_posq_with_weights, dim3((padded_atoms + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                       (const float4*) posq_d, weight_d, atom_of_slot_d, padded_atoms, (float4*) dst_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_clear_trailer_flags(void* wire_d, int ranks, int slots_per_rank, int trailer_slot, void* stream) {
    if (ranks <= 0) return 0;
    hipLaunchKernelGGL(k_clear_trailer_flags, dim3((ranks + 63) / 64), dim3(64), 0, (hipStream_t) stream, (uint4*) wire_d, ranks, slots_per_rank, trailer_slot);
    return (int) hipGetLastError();
}

extern "C" int ommhip_encode_wire(const void* pos_d, const int* atom_of_slot_d, int slot0, int slot1, const double box[6], void* wire_d, void* stream) {
    if (slot1 <= slot0) return 0;
    hipLaunchKernelGGL(k_encode_wire, dim3((slot1 - slot0 + 255) / 256), dim3(256), 0, (hipStream_t) stream,
                       (const double4*) pos_d, atom_of_slot_d, slot0, slot1, 1.0 / box[0], 1.0 / box[2], 1.0 / box[5], box[1], box[3], box[4], (uint4*) wire_d);
    return (int) hipGetLastError();
}
