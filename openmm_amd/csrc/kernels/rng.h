// Counter-based random numbers shared by the integration kernels (integrate.hip, custom_integrator.hip).
#ifndef OMMHIP_RNG_H_
#define OMMHIP_RNG_H_
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 (Salmon et al., SC'11).  counter = (atom, stepLo, stepHi, stream),
// key = seed.  Stateless, so no RNG pool has to be stored, reordered or checkpointed.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    unsigned hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    unsigned hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    unsigned n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void philox4x32(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// three independent N(0,1) deviates
__device__ __forceinline__ double3 gaussian3(unsigned atom, unsigned long long step, unsigned long long seed) {
    unsigned c[4] = {atom, (unsigned) step, (unsigned) (step >> 32), 0x4f4d4d48u};
    philox4x32(c, (unsigned) seed, (unsigned) (seed >> 32));
    // Box-Muller in single precision (the deviates only feed the thermostat; the reference GPU platforms draw their
    // normals in float as well), accumulated into the double-precision velocities by the caller.
    const float inv32 = 1.0f / 4294967296.0f;
    const float u0 = fmaxf(((float) c[0] + 0.5f) * inv32, 1.0e-10f), u1 = ((float) c[1]) * inv32;
    const float u2 = fmaxf(((float) c[2] + 0.5f) * inv32, 1.0e-10f), u3 = ((float) c[3]) * inv32;
#ifndef OMMHIP_EMU
    // The hardware's own transcendentals: v_log_f32 (base 2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in REVOLUTIONS: u itself, no
    // multiplication by 2 pi and no range reduction) -- a dozen instructions instead of the ~250 of logf / sincosf with their argument
    // reduction, three times per water molecule in a kernel that is one wavefront per SIMD deep.  Absolute error of a deviate ~1e-6.
    const float r0 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0)), r1 = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u2));
    const float s0 = __builtin_amdgcn_sinf(u1), c0 = __builtin_amdgcn_cosf(u1), c1 = __builtin_amdgcn_cosf(u3);
#else
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    float s0, c0, s1, c1;
    sincosf(6.2831853071795865f * u1, &s0, &c0);
    sincosf(6.2831853071795865f * u3, &s1, &c1);
    (void) s1;
#endif
    return make_double3(r0 * c0, r0 * s0, r1 * c1);
}

// four uniform deviates in [0, 1) for (atom, draw): an independent stream of the same generator
__device__ __forceinline__ void uniform4(unsigned atom, unsigned long long draw, unsigned long long seed, float (&u)[4]) {
    unsigned c[4] = {atom, (unsigned) draw, (unsigned) (draw >> 32), 0x554e4946u};
    philox4x32(c, (unsigned) seed, (unsigned) (seed >> 32));
    const float inv32 = 1.0f / 4294967296.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) u[i] = fminf((float) c[i] * inv32, 0.99999994f);
}

}  // namespace

#endif
