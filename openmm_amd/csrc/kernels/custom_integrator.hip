// CustomIntegrator's per-degree-of-freedom computations on the device: a small stack machine that interprets the postfix form of a
// Lepton expression, one thread per atom (its three degrees of freedom in turn), double precision.  See include/openmm_hip_kernels.h
// (ommhip_vm_*) for the contract; Reference: ReferenceCustomDynamics.cpp:357-380 (computePerDof), :300-320 (ComputeSum).
//
// An integrator step is a handful of such computations of a few operations each over ~100 bytes of state per atom: the work is launch- and
// bandwidth-bound, so consecutive computations share a launch and keep the atom's state in registers between them (x, v and the per-DOF
// variables are re-read only from what the thread itself wrote).
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

__device__ __forceinline__ double vm_run(const ommhip_vm_state& s, const ommhip_vm_step& st, const DofVars& var, int atom, int axis) {
    double stack[OMMHIP_VM_STACK];
    int top = -1;                      // index of the top of the stack
    for (int pc = st.first; pc < st.first + st.count; pc++) {
        const ommhip_vm_instruction in = s.program[pc];
        switch (in.op) {
            case OMMHIP_VM_CONSTANT: stack[++top] = in.value; break;
            case OMMHIP_VM_VARIABLE:
                switch (in.arg) {
                    case 0: stack[++top] = var.x; break;
                    case 1: stack[++top] = var.v; break;
                    case 2: stack[++top] = var.f; break;
                    case 3: stack[++top] = var.m; break;
                    case 4: stack[++top] = var.gaussian; break;
                    case 5: stack[++top] = var.uniform; break;
                    default: stack[++top] = s.per_dof[(size_t) (in.arg - 6) * 3 * s.num_atoms + 3 * atom + axis]; break;
                }
                break;
            case OMMHIP_VM_GLOBAL: stack[++top] = s.globals[in.arg]; break;
            case OMMHIP_VM_ADD: top--; stack[top] = stack[top] + stack[top + 1]; break;
            case OMMHIP_VM_SUBTRACT: top--; stack[top] = stack[top] - stack[top + 1]; break;
            case OMMHIP_VM_MULTIPLY: top--; stack[top] = stack[top] * stack[top + 1]; break;
            case OMMHIP_VM_DIVIDE: top--; stack[top] = stack[top] / stack[top + 1]; break;
            case OMMHIP_VM_POWER: top--; stack[top] = pow(stack[top], stack[top + 1]); break;
            case OMMHIP_VM_NEGATE: stack[top] = -stack[top]; break;
            case OMMHIP_VM_SQRT: stack[top] = sqrt(stack[top]); break;
            case OMMHIP_VM_EXP: stack[top] = exp(stack[top]); break;
            case OMMHIP_VM_LOG: stack[top] = log(stack[top]); break;
            case OMMHIP_VM_SIN: stack[top] = sin(stack[top]); break;
            case OMMHIP_VM_COS: stack[top] = cos(stack[top]); break;
            case OMMHIP_VM_SEC: stack[top] = 1.0 / cos(stack[top]); break;
            case OMMHIP_VM_CSC: stack[top] = 1.0 / sin(stack[top]); break;
            case OMMHIP_VM_TAN: stack[top] = tan(stack[top]); break;
            case OMMHIP_VM_COT: stack[top] = 1.0 / tan(stack[top]); break;
            case OMMHIP_VM_ASIN: stack[top] = asin(stack[top]); break;
            case OMMHIP_VM_ACOS: stack[top] = acos(stack[top]); break;
            case OMMHIP_VM_ATAN: stack[top] = atan(stack[top]); break;
            case OMMHIP_VM_ATAN2: top--; stack[top] = atan2(stack[top], stack[top + 1]); break;
            case OMMHIP_VM_SINH: stack[top] = sinh(stack[top]); break;
            case OMMHIP_VM_COSH: stack[top] = cosh(stack[top]); break;
            case OMMHIP_VM_TANH: stack[top] = tanh(stack[top]); break;
            case OMMHIP_VM_ERF: stack[top] = erf(stack[top]); break;
            case OMMHIP_VM_ERFC: stack[top] = erfc(stack[top]); break;
            case OMMHIP_VM_STEP: stack[top] = stack[top] >= 0.0 ? 1.0 : 0.0; break;
            case OMMHIP_VM_DELTA: stack[top] = stack[top] == 0.0 ? 1.0 : 0.0; break;
            case OMMHIP_VM_SQUARE: stack[top] = stack[top] * stack[top]; break;
            case OMMHIP_VM_CUBE: stack[top] = stack[top] * stack[top] * stack[top]; break;
            case OMMHIP_VM_RECIPROCAL: stack[top] = 1.0 / stack[top]; break;
            case OMMHIP_VM_ADD_CONSTANT: stack[top] = stack[top] + in.value; break;
            case OMMHIP_VM_MULTIPLY_CONSTANT: stack[top] = stack[top] * in.value; break;
            case OMMHIP_VM_POWER_CONSTANT: stack[top] = pow(stack[top], in.value); break;
            case OMMHIP_VM_MIN: top--; stack[top] = fmin(stack[top], stack[top + 1]); break;
            case OMMHIP_VM_MAX: top--; stack[top] = fmax(stack[top], stack[top + 1]); break;
            case OMMHIP_VM_ABS: stack[top] = fabs(stack[top]); break;
            case OMMHIP_VM_FLOOR: stack[top] = floor(stack[top]); break;
            case OMMHIP_VM_CEIL: stack[top] = ceil(stack[top]); break;
            case OMMHIP_VM_SELECT: top -= 2; stack[top] = stack[top] != 0.0 ? stack[top + 1] : stack[top + 2]; break;
            default: break;
        }
    }
    return stack[0];
}

__global__ __launch_bounds__(128) void k_vm_per_dof(VmArgs a) {
    __shared__ double partial[2];
    const ommhip_vm_state& s = a.s;
    const int atom = blockIdx.x * 128 + threadIdx.x;
    double sum = 0;
    if (atom < s.num_atoms) {
        double4* pos = (double4*) s.pos;
        double4* vel = (double4*) s.vel;
        double4 x = pos[atom], v = vel[atom];
        if (v.w != 0.0) {
            bool xChanged = false, vChanged = false;
            for (int i = 0; i < a.numSteps; i++) {
                const ommhip_vm_step& st = a.step[i];
                double g[3] = {0, 0, 0};
                float u[4] = {0, 0, 0, 0};
                if (st.uses_random & 1) { const double3 n = gaussian3((unsigned) atom, st.draw, s.seed); g[0] = n.x; g[1] = n.y; g[2] = n.z; }
                if (st.uses_random & 2) uniform4((unsigned) atom, st.draw, s.seed, u);
                double result[3];
#pragma unroll
                for (int axis = 0; axis < 3; axis++) {
                    DofVars var;
                    var.x = axis == 0 ? x.x : (axis == 1 ? x.y : x.z);
                    var.v = axis == 0 ? v.x : (axis == 1 ? v.y : v.z);
                    var.f = st.force != nullptr ? st.force[3 * (size_t) atom + axis] : 0.0;
                    var.m = 1.0 / v.w;
                    var.gaussian = g[axis];
                    var.uniform = (double) u[axis];
                    result[axis] = vm_run(s, st, var, atom, axis);
                }
                if (st.target == 0) { x.x = result[0]; x.y = result[1]; x.z = result[2]; xChanged = true; }
                else if (st.target == 1) { v.x = result[0]; v.y = result[1]; v.z = result[2]; vChanged = true; }
                else if (st.target >= 2) {
                    double* out = s.per_dof + (size_t) (st.target - 2) * 3 * s.num_atoms + 3 * (size_t) atom;
                    out[0] = result[0]; out[1] = result[1]; out[2] = result[2];
                }
                else sum += result[0] + result[1] + result[2];
            }
            if (xChanged) pos[atom] = x;
            if (vChanged) vel[atom] = v;
        }
    }
    if (a.numSteps == 1 && a.step[0].target < 0) {
        sum = wave_sum(sum);
        if ((threadIdx.x & 63) == 0) partial[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) s.sum_scratch[blockIdx.x] = partial[0] + partial[1];
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

}  // namespace

extern "C" int ommhip_vm_per_dof(const ommhip_vm_state* state, int num_steps, const ommhip_vm_step* steps, void* stream) {
    if (num_steps <= 0) return 0;
    if (num_steps > OMMHIP_VM_MAX_STEPS || state->num_atoms <= 0) return 1;
    VmArgs a;
    a.s = *state; a.numSteps = num_steps;
    bool sums = false;
    for (int i = 0; i < num_steps; i++) { a.step[i] = steps[i]; sums = sums || steps[i].target < 0; }
    for (int i = num_steps; i < OMMHIP_VM_MAX_STEPS; i++) a.step[i] = steps[0];
    const int blocks = (state->num_atoms + 127) / 128;
    if (sums && (num_steps != 1 || blocks > OMMHIP_KE_SCRATCH * 64 || state->sum_scratch == NULL || state->sum_result == NULL)) return 1;
    hipLaunchKernelGGL(k_vm_per_dof, dim3(blocks), dim3(128), 0, (hipStream_t) stream, a);
    if (sums) hipLaunchKernelGGL(k_vm_sum, dim3(1), dim3(64), 0, (hipStream_t) stream, (const double*) state->sum_scratch, blocks, state->sum_result);
    return (int) hipGetLastError();
}

extern "C" int ommhip_forces_to_atom_order(const long long* force_d, const int* slot_of_atom_d, int num_atoms, int padded_atoms, double* out_d, void* stream) {
    if (num_atoms <= 0) return 0;
    hipLaunchKernelGGL(k_forces_to_atom_order, dim3((num_atoms + 255) / 256), dim3(256), 0, (hipStream_t) stream, (const omm_fixed*) force_d, slot_of_atom_d, num_atoms, padded_atoms, out_d);
    return (int) hipGetLastError();
}
