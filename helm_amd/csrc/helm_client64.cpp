// helm_client64.cpp — CPU key generation, encryption, decryption for the shortint
// (LUT / arithmetic mode) path: 64-bit torus, ciphertexts under the big key
// (include/helm_client.h).  Client-side counterpart of tfhe::shortint::{gen_keys,
// ClientKey} as HELM uses them (reference src/bin/helm.rs:301, src/circuit.rs:982-996,1092).
#include "../../include/helm_client.h"
#include "rng.hpp"

#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err64;
int fail64(int code, const std::string &m)
{
    g_err64 = m;
    return code;
}

using Rng64 = helm_rng::Rng;

} // namespace

struct helm_si_client_key {
    helm_si_params P;
    double lwe_std, glwe_std;
    std::vector<uint64_t> lwe_sk, glwe_sk, bsk, ksk;
    uint64_t delta;
    Rng64 enc_rng;
};

extern "C" {

int helm_si_client_named_params(const char *name, helm_si_params *p, double *lwe_std, double *glwe_std)
{
    if (!name || !p || !lwe_std || !glwe_std) return fail64(HELM_ERR_INVALID, "null argument");
    std::memset(p, 0, sizeof(*p));
    const std::string s(name);
    p->k = 1;
    p->message_modulus = 4;
    p->carry_modulus = 4;
    if (s == "shortint_m2c2") {
        // tfhe 0.4 shortint PARAM_MESSAGE_2_CARRY_2_KS_PBS [recalled, SURVEY.md App. B]
        p->n = 742; p->N = 2048; p->pbs_l = 1; p->pbs_logB = 23; p->ks_l = 5; p->ks_logB = 3;
        *lwe_std = 0.000007069849454709433;
        *glwe_std = 0.00000000000000029403601535432533;
    } else if (s == "shortint_m2c2_multibit3") {
        // PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS, the set of reference src/bin/helm.rs:83
        // [dimensions recalled; noise: the LWE value extrapolated along tfhe's
        // security line through the n = 684 and n = 742 sets, the GLWE value of the N = 2048 sets]
        p->n = 888; p->N = 2048; p->pbs_l = 1; p->pbs_logB = 21; p->ks_l = 3; p->ks_logB = 4;
        p->grouping_factor = 3;
        *lwe_std = 0.00000049;
        *glwe_std = 0.00000000000000029403601535432533;
    } else if (s == "si_toy_2048_mb3") {
        p->n = 12; p->N = 2048; p->pbs_l = 1; p->pbs_logB = 21; p->ks_l = 3; p->ks_logB = 4;
        p->grouping_factor = 3;
        *lwe_std = 1e-9;
        *glwe_std = 1e-16;
    } else if (s == "shortint_m1c1") {
        // tfhe 0.4 shortint PARAM_MESSAGE_1_CARRY_1_KS_PBS, the set the reference binary installs for LUT mode
        // (src/bin/helm.rs:301) [dimensions recalled, SURVEY.md App. B: n = 684, k = 3, N = 512, PBS 18 x 1,
        // KS 4 x 3, message_modulus = carry_modulus = 2; LWE noise recalled; GLWE noise interpolated along tfhe's
        // security line between its k N = 1024 (4.99e-8) and k N = 2048 (2.94e-16) values: an approximate set]
        p->n = 684; p->k = 3; p->N = 512; p->pbs_l = 1; p->pbs_logB = 18; p->ks_l = 3; p->ks_logB = 4;
        p->message_modulus = 2; p->carry_modulus = 2;
        *lwe_std = 0.00002043357207216175;
        *glwe_std = 0.0000000000038;
    } else if (s == "si_toy_512_k3") { // the k = 3 kernel at toy size (oracle-sized; 16 plaintext values like the other toys)
        p->n = 10; p->k = 3; p->N = 512; p->pbs_l = 1; p->pbs_logB = 18; p->ks_l = 3; p->ks_logB = 4;
        *lwe_std = 1e-9;
        *glwe_std = 1e-15;
    } else if (s == "si_toy_512_k2") {
        p->n = 9; p->k = 2; p->N = 512; p->pbs_l = 1; p->pbs_logB = 20; p->ks_l = 4; p->ks_logB = 3;
        *lwe_std = 1e-9;
        *glwe_std = 1e-15;
    } else if (s == "si_toy_1024_mb2") {
        p->n = 8; p->N = 1024; p->pbs_l = 1; p->pbs_logB = 22; p->ks_l = 5; p->ks_logB = 3;
        p->grouping_factor = 2;
        *lwe_std = 1e-9;
        *glwe_std = 1e-15;
    } else if (s == "si_toy_512") {
        p->n = 12; p->N = 512; p->pbs_l = 2; p->pbs_logB = 15; p->ks_l = 4; p->ks_logB = 4;
        *lwe_std = 1e-9;
        *glwe_std = 1e-15;
    } else if (s == "si_toy_1024") {
        p->n = 10; p->N = 1024; p->pbs_l = 1; p->pbs_logB = 23; p->ks_l = 5; p->ks_logB = 3;
        *lwe_std = 1e-9;
        *glwe_std = 1e-15;
    } else if (s == "si_toy_2048") {
        p->n = 8; p->N = 2048; p->pbs_l = 1; p->pbs_logB = 23; p->ks_l = 5; p->ks_logB = 3;
        *lwe_std = 1e-9;
        *glwe_std = 1e-16;
    } else if (s == "si_toy_2048_l2") {
        p->n = 6; p->N = 2048; p->pbs_l = 2; p->pbs_logB = 14; p->ks_l = 3; p->ks_logB = 5;
        *lwe_std = 1e-9;
        *glwe_std = 1e-16;
    } else
        return fail64(HELM_ERR_INVALID, "unknown shortint parameter set '" + s + "'");
    return 0;
}

int helm_si_client_keygen(const helm_si_params *params, double lwe_std, double glwe_std, uint64_t seed,
                          helm_si_client_key **out)
{
    if (!params || !out) return fail64(HELM_ERR_INVALID, "null argument");
    *out = nullptr;
    const helm_si_params &P = *params;
    const int t = P.message_modulus * P.carry_modulus;
    if (P.n < 1 || P.k < 1 || P.N < 2 || (P.N & (P.N - 1)) || P.pbs_l < 1 || P.ks_l < 1 || P.pbs_logB < 1 ||
        P.ks_logB < 1 || P.pbs_logB * P.pbs_l > 64 || P.ks_logB * P.ks_l > 64 || t < 2 || (t & (t - 1)))
        return fail64(HELM_ERR_INVALID, "bad parameter set");
    const int g = P.grouping_factor > 1 ? P.grouping_factor : 1;
    if (g > 4 || P.n % g) return fail64(HELM_ERR_INVALID, "grouping_factor must be at most 4 and divide n");
    helm_si_client_key *K = new (std::nothrow) helm_si_client_key();
    if (!K) return fail64(HELM_ERR_OOM, "key");
    // seed 0: ChaCha20 streams under an OS-drawn key; otherwise the deterministic test generator (rng.hpp)
    std::unique_ptr<helm_rng::Source> src_p;
    try {
        src_p.reset(new helm_rng::Source(seed));
        K->enc_rng = src_p->encryption(0xE1C64);
    } catch (const std::exception &e) {
        delete K;
        return fail64(HELM_ERR_STATE, e.what());
    }
    const helm_rng::Source &src = *src_p;
    K->P = P;
    K->lwe_std = lwe_std;
    K->glwe_std = glwe_std;
    K->delta = (1ull << 63) / (uint64_t)t;
    const int n = P.n, k = P.k, N = P.N, k1 = k + 1, kN = k * N;
    Rng64 r0 = src.stream(0x51);
    K->lwe_sk.resize(n);
    for (auto &b : K->lwe_sk) b = r0.next() >> 63;
    K->glwe_sk.resize(kN);
    for (auto &b : K->glwe_sk) b = r0.next() >> 63;

    // ---- bootstrapping key, [n_ggsw][l][k+1 rows][k+1 polys][N] ---------------------------
    //      classical: GGSW(s_i) for every mask word; multi-bit: per group of g words the 2^g
    //      GGSWs of the subset indicators (bit i of the subset index = member i of the group)
    const size_t poly_per_i = (size_t)P.pbs_l * k1 * k1;
    const int subsets = g > 1 ? 1 << g : 1;
    const int n_ggsw = g > 1 ? (n / g) * subsets : n;
    K->bsk.assign((size_t)n_ggsw * poly_per_i * N, 0);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n_ggsw; i++) {
        uint64_t message = 0;
        if (g > 1) {
            const int grp = i / subsets, S = i % subsets;
            message = 1;
            for (int q = 0; q < g; q++) message &= ((S >> q) & 1) ? K->lwe_sk[grp * g + q] : 1 - K->lwe_sk[grp * g + q];
        } else
            message = K->lwe_sk[i];
        Rng64 r = src.stream(0x6400000 + (uint64_t)i);
        std::vector<uint64_t> body(N);
        for (int j = 0; j < P.pbs_l; j++)
            for (int row = 0; row < k1; row++) {
                uint64_t *glwe = K->bsk.data() + (((size_t)i * P.pbs_l + j) * k1 + row) * k1 * N;
                for (int u = 0; u < N; u++) body[u] = r.noise64(glwe_std);
                for (int c = 0; c < k; c++) {
                    uint64_t *A = glwe + (size_t)c * N;
                    for (int u = 0; u < N; u++) A[u] = r.next();
                    const uint64_t *S = K->glwe_sk.data() + (size_t)c * N;
                    for (int u = 0; u < N; u++) {
                        if (!S[u]) continue;
                        for (int v = 0; v < N - u; v++) body[v + u] += A[v];
                        for (int v = N - u; v < N; v++) body[v + u - N] -= A[v];
                    }
                }
                std::memcpy(glwe + (size_t)k * N, body.data(), sizeof(uint64_t) * N);
                if (message) glwe[(size_t)row * N] += (uint64_t)1 << (64 - P.pbs_logB * (j + 1));
            }
    }
    // ---- keyswitching key: [k*N][ks_l][n+1] ---------------------------------------------
    K->ksk.assign((size_t)kN * P.ks_l * (n + 1), 0);
    #pragma omp parallel for schedule(dynamic, 16)
    for (int u = 0; u < kN; u++) {
        Rng64 r = src.stream(0x6500000 + (uint64_t)u);
        for (int j = 0; j < P.ks_l; j++) {
            uint64_t *ct = K->ksk.data() + ((size_t)u * P.ks_l + j) * (n + 1);
            uint64_t b = r.noise64(lwe_std);
            for (int i = 0; i < n; i++) {
                ct[i] = r.next();
                if (K->lwe_sk[i]) b += ct[i];
            }
            if (K->glwe_sk[u]) b += (uint64_t)1 << (64 - P.ks_logB * (j + 1));
            ct[n] = b;
        }
    }
    *out = K;
    return 0;
}

void helm_si_client_key_free(helm_si_client_key *key) { delete key; }

int helm_si_client_params(const helm_si_client_key *key, helm_si_params *out)
{
    if (!key || !out) return fail64(HELM_ERR_INVALID, "null argument");
    *out = key->P;
    return 0;
}
int helm_si_client_noise(const helm_si_client_key *key, double *lwe_noise_std, double *glwe_noise_std)
{
    if (!key || !lwe_noise_std || !glwe_noise_std) return fail64(HELM_ERR_INVALID, "null argument");
    *lwe_noise_std = key->lwe_std;
    *glwe_noise_std = key->glwe_std;
    return 0;
}

size_t helm_si_client_bsk_words(const helm_si_client_key *key) { return key ? key->bsk.size() : 0; }
size_t helm_si_client_ksk_words(const helm_si_client_key *key) { return key ? key->ksk.size() : 0; }
const uint64_t *helm_si_client_bsk(const helm_si_client_key *key) { return key ? key->bsk.data() : nullptr; }
const uint64_t *helm_si_client_ksk(const helm_si_client_key *key) { return key ? key->ksk.data() : nullptr; }
const uint64_t *helm_si_client_lwe_secret(const helm_si_client_key *key) { return key ? key->lwe_sk.data() : nullptr; }
const uint64_t *helm_si_client_glwe_secret(const helm_si_client_key *key) { return key ? key->glwe_sk.data() : nullptr; }

int helm_si_client_encrypt(helm_si_client_key *key, const uint64_t *values, int64_t count, uint64_t *out)
{
    if (!key || !values || !out || count < 0) return fail64(HELM_ERR_INVALID, "bad argument");
    const int dim = key->P.k * key->P.N;
    const uint64_t t = (uint64_t)key->P.message_modulus * key->P.carry_modulus;
    for (int64_t g = 0; g < count; g++) {
        uint64_t *ct = out + (size_t)g * (dim + 1);
        uint64_t b = key->enc_rng.noise64(key->glwe_std);
        for (int i = 0; i < dim; i++) {
            ct[i] = key->enc_rng.next();
            if (key->glwe_sk[i]) b += ct[i];
        }
        ct[dim] = b + (values[g] % t) * key->delta;
    }
    return 0;
}

int helm_si_client_phase(const helm_si_client_key *key, const uint64_t *lwe, int64_t count, int small, uint64_t *ph)
{
    if (!key || !lwe || !ph || count < 0) return fail64(HELM_ERR_INVALID, "bad argument");
    const int dim = small ? key->P.n : key->P.k * key->P.N;
    const uint64_t *sk = small ? key->lwe_sk.data() : key->glwe_sk.data();
    for (int64_t g = 0; g < count; g++) {
        const uint64_t *ct = lwe + (size_t)g * (dim + 1);
        uint64_t v = ct[dim];
        for (int i = 0; i < dim; i++)
            if (sk[i]) v -= ct[i];
        ph[g] = v;
    }
    return 0;
}

int helm_si_client_decrypt(const helm_si_client_key *key, const uint64_t *lwe, int64_t count, uint64_t *values)
{
    if (!key || !lwe || !values || count < 0) return fail64(HELM_ERR_INVALID, "bad argument");
    std::vector<uint64_t> ph((size_t)count);
    if (int rc = helm_si_client_phase(key, lwe, count, 0, ph.data())) return rc;
    const uint64_t t = (uint64_t)key->P.message_modulus * key->P.carry_modulus;
    for (int64_t g = 0; g < count; g++) {
        // round to the nearest multiple of delta; the padding bit wraps into mod 2t, then mod t
        const uint64_t v = (ph[(size_t)g] + key->delta / 2) / key->delta;
        values[g] = v % t;
    }
    return 0;
}

} // extern "C"
