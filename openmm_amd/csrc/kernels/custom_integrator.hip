// CustomIntegrator's per-degree-of-freedom computations on the device: a small stack machine that interprets the postfix form of a
// Lepton expression, one thread per degree of freedom, double precision, the value stack in LDS.  See include/openmm_hip_kernels.h
// (ommhip_vm_*) for the contract; Reference: ReferenceCustomDynamics.cpp:357-380 (computePerDof), :300-320 (ComputeSum).
//
// An integrator step is a handful of such computations of a few operations each over ~100 bytes of state per atom: the work is launch- and
// bandwidth-bound, so consecutive computations share a launch and keep x and v of the degree of freedom in registers between them (the
// per-DOF variables are re-read only from what the thread itself wrote).
#include "common.h"
#include "../../../include/openmm_hip_kernels.h"
#include "rng.h"

using namespace omm;

namespace {

struct VmArgs {
    ommhip_vm_state s;
    int numSteps;
    ommhip_vm_step step[OMMHIP_VM_MAX_STEPS];
};

struct DofVars {
    double x, v, f, m, gaussian, uniform;
};

#define VM_BLOCK 64
#define VM_MAX_BLOCKS 4096          // grid-stride beyond that (and the limit of the partial sums of a ComputeSum)

// the value stack of a thread: column threadIdx.x of an LDS array (dynamic indexing of a private array would live in scratch memory)
#define STACK(i) stack[(i) * VM_BLOCK + lane]

__device__ __forceinline__ double vm_run(const ommhip_vm_state& s, const ommhip_vm_step& st, const DofVars& var, size_t dof, double* stack, const int lane) {
    int top = -1;                      // index of the top of the stack
    for (int pc = st.first; pc < st.first + st.count; pc++) {
        const ommhip_vm_instruction in = s.program[pc];
        double a = top >= 0 ? STACK(top) : 0.0;
        switch (in.op) {
            case OMMHIP_VM_CONSTANT: STACK(++top) = in.value; break;
            case OMMHIP_VM_VARIABLE: {
                double value;
                switch (in.arg) {
                    case 0: value = var.x; break;
                    case 1: value = var.v; break;
                    case 2: value = var.f; break;
                    case 3: value = var.m; break;
                    case 4: value = var.gaussian; break;
                    case 5: value = var.uniform; break;
                    default: value = s.per_dof[(size_t) (in.arg - 6) * 3 * s.num_atoms + dof]; break;
                }
                STACK(++top) = value;
                break;
            }
            case OMMHIP_VM_GLOBAL: STACK(++top) = s.globals[in.arg]; break;
            case OMMHIP_VM_ADD: top--; STACK(top) = STACK(top) + a; break;
            case OMMHIP_VM_SUBTRACT: top--; STACK(top) = STACK(top) - a; break;
            case OMMHIP_VM_MULTIPLY: top--; STACK(top) = STACK(top) * a; break;
            case OMMHIP_VM_DIVIDE: top--; STACK(top) = STACK(top) / a; break;
            case OMMHIP_VM_POWER: top--; STACK(top) = pow(STACK(top), a); break;
            case OMMHIP_VM_NEGATE: STACK(top) = -a; break;
            case OMMHIP_VM_SQRT: STACK(top) = sqrt(a); break;
            case OMMHIP_VM_EXP: STACK(top) = exp(a); break;
            case OMMHIP_VM_LOG: STACK(top) = log(a); break;
            case OMMHIP_VM_SIN: STACK(top) = sin(a); break;
            case OMMHIP_VM_COS: STACK(top) = cos(a); break;
            case OMMHIP_VM_SEC: STACK(top) = 1.0 / cos(a); break;
            case OMMHIP_VM_CSC: STACK(top) = 1.0 / sin(a); break;
            case OMMHIP_VM_TAN: STACK(top) = tan(a); break;
            case OMMHIP_VM_COT: STACK(top) = 1.0 / tan(a); break;
            case OMMHIP_VM_ASIN: STACK(top) = asin(a); break;
            case OMMHIP_VM_ACOS: STACK(top) = acos(a); break;
            case OMMHIP_VM_ATAN: STACK(top) = atan(a); break;
            case OMMHIP_VM_ATAN2: top--; STACK(top) = atan2(STACK(top), a); break;
            case OMMHIP_VM_SINH: STACK(top) = sinh(a); break;
            case OMMHIP_VM_COSH: STACK(top) = cosh(a); break;
            case OMMHIP_VM_TANH: STACK(top) = tanh(a); break;
            case OMMHIP_VM_ERF: STACK(top) = erf(a); break;
            case OMMHIP_VM_ERFC: STACK(top) = erfc(a); break;
            case OMMHIP_VM_STEP: STACK(top) = a >= 0.0 ? 1.0 : 0.0; break;
            case OMMHIP_VM_DELTA: STACK(top) = a == 0.0 ? 1.0 : 0.0; break;
            case OMMHIP_VM_SQUARE: STACK(top) = a * a; break;
            case OMMHIP_VM_CUBE: STACK(top) = a * a * a; break;
            case OMMHIP_VM_RECIPROCAL: STACK(top) = 1.0 / a; break;
            case OMMHIP_VM_ADD_CONSTANT: STACK(top) = a + in.value; break;
            case OMMHIP_VM_MULTIPLY_CONSTANT: STACK(top) = a * in.value; break;
            case OMMHIP_VM_POWER_CONSTANT: STACK(top) = pow(a, in.value); break;
            case OMMHIP_VM_MIN: top--; STACK(top) = fmin(STACK(top), a); break;
            case OMMHIP_VM_MAX: top--; STACK(top) = fmax(STACK(top), a); break;
            case OMMHIP_VM_ABS: STACK(top) = fabs(a); break;
            case OMMHIP_VM_FLOOR: STACK(top) = floor(a); break;
            case OMMHIP_VM_CEIL: STACK(top) = ceil(a); break;
            case OMMHIP_VM_SELECT: top -= 2; STACK(top) = STACK(top) != 0.0 ? STACK(top + 1) : a; break;
            default: break;
        }
    }
    return STACK(0);
}

// one thread per degree of freedom (grid-stride): the steps of the launch one after the other, x and v of the thread's own degree of
// freedom in registers between them
__global__ __launch_bounds__(VM_BLOCK) void k_vm_per_dof(VmArgs a) {
    __shared__ double stack[OMMHIP_VM_STACK * VM_BLOCK];
    const ommhip_vm_state& s = a.s;
    const int lane = threadIdx.x;
    const size_t numDof = 3 * (size_t) s.num_atoms;
    double sum = 0;
    double* pos = (double*) s.pos;
    double* vel = (double*) s.vel;
    for (size_t dof = (size_t) blockIdx.x * VM_BLOCK + lane; dof < numDof; dof += (size_t) gridDim.x * VM_BLOCK) {
        const int atom = (int) (dof / 3), axis = (int) (dof - 3 * (size_t) atom);
        const double invMass = vel[4 * (size_t) atom + 3];
        if (invMass == 0.0) continue;                 // a massless particle: skipped as in the reference
        double x = pos[4 * (size_t) atom + axis], v = vel[4 * (size_t) atom + axis];
        bool xChanged = false, vChanged = false;
        for (int i = 0; i < a.numSteps; i++) {
            const ommhip_vm_step& st = a.step[i];
            DofVars var;
            var.x = x; var.v = v; var.m = 1.0 / invMass;
            var.f = st.force != nullptr ? st.force[dof] : 0.0;
            var.gaussian = 0; var.uniform = 0;
            if (st.uses_random & 1) { const double3 n = gaussian3((unsigned) atom, st.draw, s.seed); var.gaussian = axis == 0 ? n.x : (axis == 1 ? n.y : n.z); }
            if (st.uses_random & 2) { float u[4]; uniform4((unsigned) atom, st.draw, s.seed, u); var.uniform = (double) (axis == 0 ? u[0] : (axis == 1 ? u[1] : u[2])); }
            const double result = vm_run(s, st, var, dof, stack, lane);
            if (st.target == 0) { x = result; xChanged = true; }
            else if (st.target == 1) { v = result; vChanged = true; }
            else if (st.target >= 2) s.per_dof[(size_t) (st.target - 2) * numDof + dof] = result;
            else sum += result;
        }
        if (xChanged) pos[4 * (size_t) atom + axis] = x;
        if (vChanged) vel[4 * (size_t) atom + axis] = v;
    }
    if (a.numSteps == 1 && a.step[0].target < 0) {
        sum = wave_sum(sum);
        if (lane == 0) s.sum_scratch[blockIdx.x] = sum;
    }
}

// the block sums of one target -1 launch in a fixed order
__global__ __launch_bounds__(64) void k_vm_sum(const double* __restrict__ scratch, int count, double* __restrict__ result) {
    double sum = 0;
    for (int i = threadIdx.x; i < count; i += 64) sum += scratch[i];
    sum = wave_sum(sum);
    if (threadIdx.x == 0) *result = sum;
}

__global__ __launch_bounds__(256) void k_forces_to_atom_order(const omm_fixed* __restrict__ force, const int* __restrict__ slotOfAtom, int numAtoms, int paddedAtoms, double* __restrict__ out) {
    const int atom = blockIdx.x * 256 + threadIdx.x;
    if (atom >= numAtoms) return;
    const int slot = slotOfAtom[atom];
    out[3 * (size_t) atom] = from_fixed(force[slot]);
    out[3 * (size_t) atom + 1] = from_fixed(force[slot + paddedAtoms]);
    out[3 * (size_t) atom + 2] = from_fixed(force[slot + 2 * paddedAtoms]);
}


struct VmBondArgs {
    ommhip_vm_bonds b;
    const double4* pos; const int* slotOfAtom; int paddedAtoms;
    omm_fixed* force; double* energyBuffer; int energySlots, includeEnergy;
};

// CustomBondForce with an arbitrary expression: one thread per bond, the derivative program gives dE/dr, the energy program E
__global__ __launch_bounds__(VM_BLOCK) void k_vm_bonds(VmBondArgs a) {
    __shared__ double stack[OMMHIP_VM_STACK * VM_BLOCK];
    const int lane = threadIdx.x;
    // (the per-bond parameters are laid out as per-DOF variables are: parameter k of bond i at per_dof[k * 3 * num_atoms + i])
    ommhip_vm_state s;
    s.num_atoms = a.b.param_stride / 3; s.num_per_dof = a.b.num_params; s.per_dof = const_cast<double*>(a.b.params);
    s.globals = a.b.globals; s.program = a.b.program;
    ommhip_vm_step energyStep = {a.b.energy_first, a.b.energy_count, 0, 0, nullptr, 0}, derivStep = {a.b.deriv_first, a.b.deriv_count, 0, 0, nullptr, 0};
    double energy = 0;
    for (int bond = blockIdx.x * VM_BLOCK + lane; bond < a.b.num_bonds; bond += gridDim.x * VM_BLOCK) {
        const int i = a.b.atoms[2 * bond], j = a.b.atoms[2 * bond + 1];
        const double4 pi = a.pos[i], pj = a.pos[j];
        double dx = pj.x - pi.x, dy = pj.y - pi.y, dz = pj.z - pi.z;
        if (a.b.periodic) {
            // ReferenceForce::getDeltaRPeriodicTriclinic
            double n = floor(dz / a.b.box[5] + 0.5); dx -= n * a.b.box[3]; dy -= n * a.b.box[4]; dz -= n * a.b.box[5];
            n = floor(dy / a.b.box[2] + 0.5); dx -= n * a.b.box[1]; dy -= n * a.b.box[2];
            n = floor(dx / a.b.box[0] + 0.5); dx -= n * a.b.box[0];
        }
        const double r = sqrt(dx * dx + dy * dy + dz * dz);
        DofVars var;
        var.x = r; var.v = 0; var.f = 0; var.m = 0; var.gaussian = 0; var.uniform = 0;
        const double dEdR = vm_run(s, derivStep, var, (size_t) bond, stack, lane);
        if (a.includeEnergy) energy += vm_run(s, energyStep, var, (size_t) bond, stack, lane);
        const double scale = r > 0 ? dEdR / r : 0.0;          // ReferenceCustomBondIxn.cpp:98-99
        add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], scale * dx, scale * dy, scale * dz);
        add_force(a.force, a.paddedAtoms, a.slotOfAtom[j], -scale * dx, -scale * dy, -scale * dz);
    }
    if (a.includeEnergy) {
        energy = wave_sum(energy);
        if (lane == 0) atomicAdd(&a.energyBuffer[blockIdx.x % a.energySlots], energy);
    }
}

__device__ __forceinline__ void vm_min_image(const ommhip_vm_bonds& b, double& dx, double& dy, double& dz) {
    if (!b.periodic) return;
    double n = floor(dz / b.box[5] + 0.5); dx -= n * b.box[3]; dy -= n * b.box[4]; dz -= n * b.box[5];
    n = floor(dy / b.box[2] + 0.5); dx -= n * b.box[1]; dy -= n * b.box[2];
    n = floor(dx / b.box[0] + 0.5); dx -= n * b.box[0];
}

// CustomAngleForce with an arbitrary expression of theta: one thread per angle
__global__ __launch_bounds__(VM_BLOCK) void k_vm_angles(VmBondArgs a) {
    __shared__ double stack[OMMHIP_VM_STACK * VM_BLOCK];
    const int lane = threadIdx.x;
    ommhip_vm_state s;
    s.num_atoms = a.b.param_stride / 3; s.num_per_dof = a.b.num_params; s.per_dof = const_cast<double*>(a.b.params);
    s.globals = a.b.globals; s.program = a.b.program;
    ommhip_vm_step energyStep = {a.b.energy_first, a.b.energy_count, 0, 0, nullptr, 0}, derivStep = {a.b.deriv_first, a.b.deriv_count, 0, 0, nullptr, 0};
    double energy = 0;
    for (int t = blockIdx.x * VM_BLOCK + lane; t < a.b.num_bonds; t += gridDim.x * VM_BLOCK) {
        const int i = a.b.atoms[3 * t], j = a.b.atoms[3 * t + 1], k = a.b.atoms[3 * t + 2];
        const double4 pi = a.pos[i], pj = a.pos[j], pk = a.pos[k];
        double ux = pi.x - pj.x, uy = pi.y - pj.y, uz = pi.z - pj.z, wx = pk.x - pj.x, wy = pk.y - pj.y, wz = pk.z - pj.z;
        vm_min_image(a.b, ux, uy, uz); vm_min_image(a.b, wx, wy, wz);
        const double u2 = ux * ux + uy * uy + uz * uz, w2 = wx * wx + wy * wy + wz * wz;
        double c = (ux * wx + uy * wy + uz * wz) / sqrt(u2 * w2);
        c = fmin(1.0, fmax(-1.0, c));
        DofVars var;
        var.x = acos(c); var.v = 0; var.f = 0; var.m = 0; var.gaussian = 0; var.uniform = 0;
        const double dEdTheta = vm_run(s, derivStep, var, (size_t) t, stack, lane);
        if (a.includeEnergy) energy += vm_run(s, energyStep, var, (size_t) t, stack, lane);
        // ReferenceAngleBondIxn.cpp:118-146: p = u x w, forces along p x u and p x w (the reference's delta vectors are -u and -w)
        const double px = uy * wz - uz * wy, py = uz * wx - ux * wz, pz = ux * wy - uy * wx;
        const double rp = fmax(sqrt(px * px + py * py + pz * pz), 1e-6);
        const double termA = dEdTheta / (u2 * rp), termC = -dEdTheta / (w2 * rp);
        const double fax = -termA * (uy * pz - uz * py), fay = -termA * (uz * px - ux * pz), faz = -termA * (ux * py - uy * px);
        const double fcx = -termC * (wy * pz - wz * py), fcy = -termC * (wz * px - wx * pz), fcz = -termC * (wx * py - wy * px);
        add_force(a.force, a.paddedAtoms, a.slotOfAtom[i], fax, fay, faz);
        add_force(a.force, a.paddedAtoms, a.slotOfAtom[k], fcx, fcy, fcz);
        add_force(a.force, a.paddedAtoms, a.slotOfAtom[j], -(fax + fcx), -(fay + fcy), -(faz + fcz));
    }
    if (a.includeEnergy) {
        energy = wave_sum(energy);
        if (lane == 0) atomicAdd(&a.energyBuffer[blockIdx.x % a.energySlots], energy);
    }
}

}  // namespace

extern "C" int ommhip_vm_per_dof(const ommhip_vm_state* state, int num_steps, const ommhip_vm_step* steps, void* stream) {
    if (num_steps <= 0) return 0;
    if (num_steps > OMMHIP_VM_MAX_STEPS || state->num_atoms <= 0) return 1;
    VmArgs a;
    a.s = *state; a.numSteps = num_steps;
    bool sums = false;
    for (int i = 0; i < num_steps; i++) { a.step[i] = steps[i]; sums = sums || steps[i].target < 0; }
    for (int i = num_steps; i < OMMHIP_VM_MAX_STEPS; i++) a.step[i] = steps[0];
    const size_t wanted = (3 * (size_t) state->num_atoms + VM_BLOCK - 1) / VM_BLOCK;
    const int blocks = (int) (wanted < VM_MAX_BLOCKS ? wanted : VM_MAX_BLOCKS);
    if (sums && (num_steps != 1 || state->sum_scratch == NULL || state->sum_result == NULL)) return 1;
    hipLaunchKernelGGL(k_vm_per_dof, dim3(blocks), dim3(VM_BLOCK), 0, (hipStream_t) stream, a);
    if (sums) hipLaunchKernelGGL(k_vm_sum, dim3(1), dim3(64), 0, (hipStream_t) stream, (const double*) state->sum_scratch, blocks, state->sum_result);
    return (int) hipGetLastError();
}

extern "C" int ommhip_forces_to_atom_order(const long long* force_d, const int* slot_of_atom_d, int num_atoms, int padded_atoms, double* out_d, void* stream) {
    if (num_atoms <= 0) return 0;
    hipLaunchKernelGGL(k_forces_to_atom_order, dim3((num_atoms + 255) / 256), dim3(256), 0, (hipStream_t) stream, (const omm_fixed*) force_d, slot_of_atom_d, num_atoms, padded_atoms, out_d);
    return (int) hipGetLastError();
}

extern "C" int ommhip_vm_bond_forces(const ommhip_vm_bonds* bonds, const void* pos_d, const int* slot_of_atom_d, int padded_atoms, long long* force_d,
                                     double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    if (bonds->num_bonds <= 0) return 0;
    VmBondArgs a;
    a.b = *bonds;
    a.pos = (const double4*) pos_d; a.slotOfAtom = slot_of_atom_d; a.paddedAtoms = padded_atoms;
    a.force = (omm_fixed*) force_d; a.energyBuffer = energy_buffer_d; a.energySlots = energy_slots; a.includeEnergy = include_energy;
    const int blocks = min(VM_MAX_BLOCKS, (bonds->num_bonds + VM_BLOCK - 1) / VM_BLOCK);
    hipLaunchKernelGGL(k_vm_bonds, dim3(blocks), dim3(VM_BLOCK), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}

extern "C" int ommhip_vm_angle_forces(const ommhip_vm_bonds* angles, const void* pos_d, const int* slot_of_atom_d, int padded_atoms, long long* force_d,
                                      double* energy_buffer_d, int energy_slots, int include_energy, void* stream) {
    if (angles->num_bonds <= 0) return 0;
    VmBondArgs a;
    a.b = *angles;
    a.pos = (const double4*) pos_d; a.slotOfAtom = slot_of_atom_d; a.paddedAtoms = padded_atoms;
    a.force = (omm_fixed*) force_d; a.energyBuffer = energy_buffer_d; a.energySlots = energy_slots; a.includeEnergy = include_energy;
    const int blocks = min(VM_MAX_BLOCKS, (angles->num_bonds + VM_BLOCK - 1) / VM_BLOCK);
    hipLaunchKernelGGL(k_vm_angles, dim3(blocks), dim3(VM_BLOCK), 0, (hipStream_t) stream, a);
    return (int) hipGetLastError();
}
