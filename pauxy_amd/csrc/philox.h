// Device random stream shared by the field kernels (k_small.hip, k_ueg.hip).
#pragma once
#include "afq_internal.h"

// --------------------------------------------------------------------------
// Philox4x32-10 counter-based generator + Box-Muller: the device stream of
// auxiliary fields used when the host passes xi == NULL (performance mode; the
// parity mode uploads numpy's legacy MT19937 normals instead).
__device__ inline void philox_round(unsigned int &c0, unsigned int &c1, unsigned int &c2, unsigned int &c3,
                                    unsigned int k0, unsigned int k1) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned int h0 = (unsigned int)(p0 >> 32), l0 = (unsigned int)p0;
    const unsigned int h1 = (unsigned int)(p1 >> 32), l1 = (unsigned int)p1;
    const unsigned int n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0; c1 = l1; c2 = n2; c3 = l0;
}

// the two normals of element pair `pair` of launch `counter` of the stream (seed, stream)
__device__ inline void philox_normal_pair(long pair, unsigned long long seed, unsigned long long stream,
                                          unsigned long long counter, double &x0, double &x1) {
    unsigned int c0 = (unsigned int)pair, c1 = (unsigned int)(pair >> 32);
    unsigned int c2 = (unsigned int)counter, c3 = (unsigned int)(counter >> 32) ^ (unsigned int)(stream * 0x9E3779B9u);
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
    for (int rd = 0; rd < 10; ++rd) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    // two 53-bit uniforms: u1 in (0, 1], u2 in (0, 1)
    const unsigned long long a = (((unsigned long long)c0 << 32) | c1) >> 11;
    const unsigned long long b = (((unsigned long long)c2 << 32) | c3) >> 11;
    const double u1 = ((double)a + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)b + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u1));
    double s, c; sincospi(2.0 * u2, &s, &c);
    x0 = rad * c; x1 = rad * s;
}

// rng.on: the auxiliary fields are drawn inside the field kernel from the device stream instead of being read from xi
// (element e = w K + n is member e & 1 of Philox pair e >> 1), and the alive flag of the step (qmc/afqmc.py:232) is set
// there too
struct FieldRng {
    int on;
    unsigned long long seed, stream, counter;
    const double *weight;
    int *alive_out;
#ifdef AFQ_TUNING
    int dbg;            // timing ablations of fields_kernel (AFQ_FIELDS_DBG; wrong results): 1 no Philox / Box-Muller, 2 no loads of
                        // the force-bias partials, 4 no stores of xbar / xs, 8 return at once, 16 no reduction of the seven sums
#endif
};
