// helm_client.cpp — CPU key generation, encryption, decryption (include/helm_client.h).
// Client-side counterpart of tfhe::boolean::{gen_keys, ClientKey} as HELM uses them
// (reference src/bin/helm.rs:241, src/circuit.rs:463-476,558).
#include "../../include/helm_client.h"
#include "rng.hpp"

#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string &m)
{
    g_err = m;
    return code;
}

using helm_rng::Rng;

} // namespace

struct helm_client_key {
    helm_hip_params P;
    double lwe_std, glwe_std;
    std::vector<uint32_t> lwe_sk;  // n bits
    std::vector<uint32_t> glwe_sk; // k*N bits
    std::vector<uint32_t> bsk, ksk;
    Rng enc_rng;
};

extern "C" {

const char *helm_client_last_error(void) { return g_err.c_str(); }

int helm_client_named_params(const char *name, helm_hip_params *p, double *lwe_std, double *glwe_std)
{
    if (!name || !p || !lwe_std || !glwe_std) return fail(HELM_ERR_INVALID, "null argument");
    std::memset(p, 0, sizeof(*p));
    p->torus_bits = 32;
    p->pbs_order = 0;
    p->grouping_factor = 1;
    const std::string s(name);
    if (s == "boolean_default") {
        // tfhe 0.4.1 boolean::DEFAULT_PARAMETERS [recalled, SURVEY.md App. B]
        p->n = 722; p->k = 2; p->N = 512; p->pbs_l = 3; p->pbs_logB = 6; p->ks_l = 4; p->ks_logB = 3;
        *lwe_std = 0.000013071021089943935;
        *glwe_std = 0.00000004990272175010415;
    } else if (s == "helm_cuda") {
        // reference src/bin/helm.rs:141-146
        p->n = 512; p->k = 1; p->N = 1024; p->pbs_l = 3; p->pbs_logB = 7; p->ks_l = 8; p->ks_logB = 2;
        *lwe_std = 0.00000002980232238769531;
        *glwe_std = 0.00000002980232238769531;
    } else if (s == "toy") {
        // small n so that the O(N^2) schoolbook oracle finishes in well under a second
        p->n = 24; p->k = 1; p->N = 512; p->pbs_l = 2; p->pbs_logB = 8; p->ks_l = 4; p->ks_logB = 4;
        *lwe_std = 1e-7;
        *glwe_std = 1e-9;
    } else if (s == "toy_k2") {
        p->n = 20; p->k = 2; p->N = 512; p->pbs_l = 3; p->pbs_logB = 6; p->ks_l = 4; p->ks_logB = 3;
        *lwe_std = 1e-7;
        *glwe_std = 1e-9;
    } else if (s == "toy_1024") {
        p->n = 16; p->k = 1; p->N = 1024; p->pbs_l = 3; p->pbs_logB = 7; p->ks_l = 8; p->ks_logB = 2;
        *lwe_std = 1e-7;
        *glwe_std = 1e-9;
    } else if (s == "toy_1024_l2") { // N = 1024 with two decomposition levels: the L = 2 instantiations of every N = 1024 build
        p->n = 16; p->k = 1; p->N = 1024; p->pbs_l = 2; p->pbs_logB = 7; p->ks_l = 8; p->ks_logB = 2;
        *lwe_std = 1e-7;
        *glwe_std = 1e-9;
    } else
        return fail(HELM_ERR_INVALID, "unknown parameter set '" + s + "'");
    return 0;
}

int helm_client_keygen(const helm_hip_params *params, double lwe_std, double glwe_std, uint64_t seed,
                       helm_client_key **out)
{
    if (!params || !out) return fail(HELM_ERR_INVALID, "null argument");
    *out = nullptr;
    const helm_hip_params &P = *params;
    if (P.torus_bits != 32) return fail(HELM_ERR_INVALID, "only torus_bits = 32");
    if (P.n < 1 || P.k < 1 || P.N < 2 || (P.N & (P.N - 1)) || P.pbs_l < 1 || P.ks_l < 1 ||
        P.pbs_logB * P.pbs_l > 32 || P.ks_logB * P.ks_l > 32 || P.pbs_logB < 1 || P.ks_logB < 1)
        return fail(HELM_ERR_INVALID, "bad parameter set");
    helm_client_key *K = new (std::nothrow) helm_client_key();
    if (!K) return fail(HELM_ERR_OOM, "key");
    // seed 0: ChaCha20 streams under an OS-drawn key; otherwise the deterministic test generator (rng.hpp)
    std::unique_ptr<helm_rng::Source> src_p;
    try {
        src_p.reset(new helm_rng::Source(seed));
        K->enc_rng = src_p->encryption(0xE1C);
    } catch (const std::exception &e) {
        delete K;
        return fail(HELM_ERR_STATE, e.what());
    }
    const helm_rng::Source &src = *src_p;
    K->P = P;
    K->lwe_std = lwe_std;
    K->glwe_std = glwe_std;
    const int n = P.n, k = P.k, N = P.N, k1 = k + 1, kN = k * N;
    Rng r0 = src.stream(0);
    K->lwe_sk.resize(n);
    for (auto &b : K->lwe_sk) b = (uint32_t)(r0.next() >> 63);
    K->glwe_sk.resize(kN);
    for (auto &b : K->glwe_sk) b = (uint32_t)(r0.next() >> 63);

    // ---- bootstrapping key: GGSW(s_i), [n][l][k+1 rows][k+1 polys][N] -----------------
    const size_t poly_per_i = (size_t)P.pbs_l * k1 * k1;
    K->bsk.assign((size_t)n * poly_per_i * N, 0);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; i++) {
        Rng r = src.stream(0x1000 + (uint64_t)i);
        std::vector<uint32_t> body(N);
        for (int j = 0; j < P.pbs_l; j++)
            for (int row = 0; row < k1; row++) {
                uint32_t *glwe = K->bsk.data() + (((size_t)i * P.pbs_l + j) * k1 + row) * k1 * N;
                for (int t = 0; t < N; t++) body[t] = r.noise32(glwe_std);
                for (int c = 0; c < k; c++) {
                    uint32_t *A = glwe + (size_t)c * N;
                    for (int t = 0; t < N; t++) A[t] = r.u32();
                    // body += A * S_c (negacyclic, S binary)
                    const uint32_t *S = K->glwe_sk.data() + (size_t)c * N;
                    for (int u = 0; u < N; u++) {
                        if (!S[u]) continue;
                        for (int t = 0; t < N - u; t++) body[t + u] += A[t];
                        for (int t = N - u; t < N; t++) body[t + u - N] -= A[t];
                    }
                }
                std::memcpy(glwe + (size_t)k * N, body.data(), sizeof(uint32_t) * N);
                // message s_i * 2^(32 - logB*(j+1)) on polynomial `row`, coefficient 0
                if (K->lwe_sk[i]) glwe[(size_t)row * N] += (uint32_t)1 << (32 - P.pbs_logB * (j + 1));
            }
    }

    // ---- keyswitching key: [k*N][ks_l][n+1] ---------------------------------------------
    K->ksk.assign((size_t)kN * P.ks_l * (n + 1), 0);
    #pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < kN; t++) {
        Rng r = src.stream(0x100000 + (uint64_t)t);
        for (int j = 0; j < P.ks_l; j++) {
            uint32_t *ct = K->ksk.data() + ((size_t)t * P.ks_l + j) * (n + 1);
            uint32_t b = r.noise32(lwe_std);
            for (int i = 0; i < n; i++) {
                ct[i] = r.u32();
                if (K->lwe_sk[i]) b += ct[i];
            }
            if (K->glwe_sk[t]) b += (uint32_t)1 << (32 - P.ks_logB * (j + 1));
            ct[n] = b;
        }
    }
    *out = K;
    return 0;
}

void helm_client_key_free(helm_client_key *key) { delete key; }

int helm_client_rng_selftest(void)
{
    return helm_rng::Rng::selftest() ? 0 : fail(HELM_ERR_STATE, "ChaCha20 block function does not reproduce RFC 8439 2.3.2");
}

int helm_client_params(const helm_client_key *key, helm_hip_params *out)
{
    if (!key || !out) return fail(HELM_ERR_INVALID, "null argument");
    *out = key->P;
    return 0;
}
size_t helm_client_bsk_words(const helm_client_key *key) { return key ? key->bsk.size() : 0; }
size_t helm_client_ksk_words(const helm_client_key *key) { return key ? key->ksk.size() : 0; }
const uint32_t *helm_client_bsk(const helm_client_key *key) { return key ? key->bsk.data() : nullptr; }
const uint32_t *helm_client_ksk(const helm_client_key *key) { return key ? key->ksk.data() : nullptr; }
const uint32_t *helm_client_lwe_secret(const helm_client_key *key) { return key ? key->lwe_sk.data() : nullptr; }
const uint32_t *helm_client_glwe_secret(const helm_client_key *key) { return key ? key->glwe_sk.data() : nullptr; }

int helm_client_encrypt_bool(helm_client_key *key, const uint8_t *bits, int64_t count, uint32_t *out)
{
    if (!key || !bits || !out || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    const int n = key->P.n;
    for (int64_t g = 0; g < count; g++) {
        uint32_t *ct = out + (size_t)g * (n + 1);
        uint32_t b = key->enc_rng.noise32(key->lwe_std);
        for (int i = 0; i < n; i++) {
            ct[i] = key->enc_rng.u32();
            if (key->lwe_sk[i]) b += ct[i];
        }
        ct[n] = b + (bits[g] ? 0x20000000u : 0xE0000000u);
    }
    return 0;
}

int helm_client_phase(const helm_client_key *key, const uint32_t *lwe, int64_t count, int big, uint32_t *ph)
{
    if (!key || !lwe || !ph || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    const int dim = big ? key->P.k * key->P.N : key->P.n;
    const uint32_t *sk = big ? key->glwe_sk.data() : key->lwe_sk.data();
    for (int64_t g = 0; g < count; g++) {
        const uint32_t *ct = lwe + (size_t)g * (dim + 1);
        uint32_t v = ct[dim];
        for (int i = 0; i < dim; i++)
            if (sk[i]) v -= ct[i];
        ph[g] = v;
    }
    return 0;
}

int helm_client_decrypt_bool(const helm_client_key *key, const uint32_t *lwe, int64_t count, uint8_t *bits)
{
    if (!key || !lwe || !bits || count < 0) return fail(HELM_ERR_INVALID, "bad argument");
    std::vector<uint32_t> ph((size_t)count);
    if (int rc = helm_client_phase(key, lwe, count, 0, ph.data())) return rc;
    for (int64_t g = 0; g < count; g++) bits[g] = ph[(size_t)g] < 0x80000000u; // circuit.rs:948
    return 0;
}

} // extern "C"
