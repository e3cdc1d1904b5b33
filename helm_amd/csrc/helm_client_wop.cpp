// helm_client_wop.cpp — CPU key generation for the WoP-PBS wide-LUT path (include/helm_wopbs.h, client section).
// Client-side counterpart of tfhe::shortint::wopbs::WopbsKey::new_wopbs_key as HELM's high_precision_lut() expects it
// (reference src/gates.rs:787-815: a WopbsKey next to the shortint ServerKey): a second secret-key pair under the
// WoP-side parameters, its bootstrapping and keyswitching keys, the two keyswitching keys between the parameter sets and
// the k+1 private functional packing keyswitching keys of the circuit bootstrap.
#include "../../include/helm_wopbs.h"
#include "rng.hpp"

#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_errw;
int failw(int code, const std::string &m)
{
    g_errw = m;
    return code;
}

using Rng = helm_rng::Rng;

// body += A * S (negacyclic, S binary)
void add_mask_times_key(const uint64_t *A, const uint64_t *S, int N, uint64_t *body)
{
    for (int u = 0; u < N; u++) {
        if (!S[u]) continue;
        for (int v = 0; v < N - u; v++) body[v + u] += A[v];
        for (int v = N - u; v < N; v++) body[v + u - N] -= A[v];
    }
}

// GLWE encryption of `message` (N words, added to the body): k mask polynomials, then the body
void glwe_encrypt(const std::vector<uint64_t> &glwe_sk, int k, int N, double std_dev, Rng &r, const uint64_t *message,
                  uint64_t *out)
{
    uint64_t *body = out + (size_t)k * N;
    for (int u = 0; u < N; u++) body[u] = r.noise64(std_dev) + (message ? message[u] : 0);
    for (int c = 0; c < k; c++) {
        uint64_t *A = out + (size_t)c * N;
        for (int u = 0; u < N; u++) A[u] = r.next();
        add_mask_times_key(A, glwe_sk.data() + (size_t)c * N, N, body);
    }
}

// LWE keyswitching key from `from` (bits) to `to` (bits): [from.size()][l][to.size()+1]
void make_ksk(const std::vector<uint64_t> &from, const std::vector<uint64_t> &to, int l, int logB, double std_dev,
              const helm_rng::Source &src, uint64_t stream0, std::vector<uint64_t> &out)
{
    const size_t n = to.size();
    out.assign(from.size() * l * (n + 1), 0);
    #pragma omp parallel for schedule(dynamic, 16)
    for (size_t u = 0; u < from.size(); u++) {
        Rng r = src.stream(stream0 + u);
        for (int j = 0; j < l; j++) {
            uint64_t *ct = out.data() + (u * l + j) * (n + 1);
            uint64_t b = r.noise64(std_dev);
            for (size_t i = 0; i < n; i++) {
                ct[i] = r.next();
                if (to[i]) b += ct[i];
            }
            if (from[u]) b += (uint64_t)1 << (64 - logB * (j + 1));
            ct[n] = b;
        }
    }
}

} // namespace

struct helm_wop_client_key {
    helm_wop_params P;
    std::vector<uint64_t> lwe_sk, glwe_sk, bsk, ksk, to_wop, to_pbs, pfpksk;
};

extern "C" {

int helm_wop_client_named_params(const char *name, helm_wop_params *p, double *lwe_std, double *glwe_std)
{
    if (!name || !p || !lwe_std || !glwe_std) return failw(HELM_ERR_INVALID, "null argument");
    std::memset(p, 0, sizeof(*p));
    const std::string s(name);
    p->k = 1;
    p->N = 2048;
    p->pbs_l = 2; p->pbs_logB = 15;
    p->ks_l = 5; p->ks_logB = 2;
    p->pfks_l = 2; p->pfks_logB = 15;
    p->cbs_l = 3; p->cbs_logB = 5;
    if (s == "wopbs_m1c1") {
        // tfhe 0.4 shortint WOPBS_PARAM_MESSAGE_1_CARRY_1_KS_PBS [dimensions and noise recalled]
        p->n = 653; p->message_modulus = 2; p->carry_modulus = 2;
        *lwe_std = 0.00003604499526942373;
        *glwe_std = 0.00000000000000029403601535432533;
    } else if (s == "wopbs_m2c2") {
        // WOPBS_PARAM_MESSAGE_2_CARRY_2_KS_PBS [dimensions and noise recalled]
        p->n = 769; p->message_modulus = 4; p->carry_modulus = 4;
        *lwe_std = 0.0000043131554647504185;
        *glwe_std = 0.00000000000000029403601535432533;
    } else if (s == "wop_toy_512") {
        p->n = 8; p->N = 512; p->ks_l = 4; p->ks_logB = 4;
        p->message_modulus = 4; p->carry_modulus = 4;
        *lwe_std = 1e-10;
        *glwe_std = 1e-16;
    } else if (s == "wop_toy_1024") {
        p->n = 10; p->N = 1024; p->ks_l = 4; p->ks_logB = 4; p->cbs_l = 2; p->cbs_logB = 8;
        p->message_modulus = 4; p->carry_modulus = 4;
        *lwe_std = 1e-10;
        *glwe_std = 1e-16;
    } else if (s == "wop_toy_2048") {
        p->n = 6; p->ks_l = 4; p->ks_logB = 4;
        p->message_modulus = 2; p->carry_modulus = 2;
        *lwe_std = 1e-10;
        *glwe_std = 1e-16;
    } else
        return failw(HELM_ERR_INVALID, "unknown WoP-PBS parameter set '" + s + "'");
    return 0;
}

int helm_wop_client_keygen(const helm_si_client_key *pbs_key, const helm_wop_params *params, double lwe_std,
                           double glwe_std, uint64_t seed, helm_wop_client_key **out)
{
    if (!pbs_key || !params || !out) return failw(HELM_ERR_INVALID, "null argument");
    *out = nullptr;
    const helm_wop_params &P = *params;
    if (P.n < 1 || P.k < 1 || P.N < 2 || (P.N & (P.N - 1)) || P.pbs_l < 1 || P.ks_l < 1 || P.pfks_l < 1 || P.cbs_l < 1 ||
        P.pbs_logB < 1 || P.ks_logB < 1 || P.pfks_logB < 1 || P.cbs_logB < 1 || P.pbs_logB * P.pbs_l > 64 ||
        P.ks_logB * P.ks_l > 64 || P.pfks_logB * P.pfks_l > 64 || P.cbs_logB * P.cbs_l > 63)
        return failw(HELM_ERR_INVALID, "bad WoP-PBS parameter set");
    helm_si_params S;
    if (helm_si_client_params(pbs_key, &S)) return failw(HELM_ERR_INVALID, "bad PBS-side key");
    if (S.message_modulus * S.carry_modulus != P.message_modulus * P.carry_modulus)
        return failw(HELM_ERR_INVALID, "the two parameter sets must share message_modulus * carry_modulus (one encoding)");
    double s_lwe_std = 0, s_glwe_std = 0;
    helm_si_client_noise(pbs_key, &s_lwe_std, &s_glwe_std);
    std::unique_ptr<helm_wop_client_key> K(new (std::nothrow) helm_wop_client_key());
    if (!K) return failw(HELM_ERR_OOM, "key");
    std::unique_ptr<helm_rng::Source> src_p;
    try {
        src_p.reset(new helm_rng::Source(seed));
    } catch (const std::exception &e) {
        return failw(HELM_ERR_STATE, e.what());
    }
    const helm_rng::Source &src = *src_p;
    K->P = P;
    const int n = P.n, k = P.k, N = P.N, k1 = k + 1, kN = k * N;
    Rng r0 = src.stream(0x77);
    K->lwe_sk.resize(n);
    for (auto &b : K->lwe_sk) b = r0.next() >> 63;
    K->glwe_sk.resize(kN);
    for (auto &b : K->glwe_sk) b = r0.next() >> 63;

    // ---- bootstrapping key: GGSW(s_i), [n][pbs_l][k+1 rows][k+1 polys][N] ----------------------------------------
    K->bsk.assign((size_t)n * P.pbs_l * k1 * k1 * N, 0);
    #pragma omp parallel for schedule(dynamic, 4)
    for (int i = 0; i < n; i++) {
        Rng r = src.stream(0x7700000 + (uint64_t)i);
        for (int j = 0; j < P.pbs_l; j++)
            for (int row = 0; row < k1; row++) {
                uint64_t *glwe = K->bsk.data() + (((size_t)i * P.pbs_l + j) * k1 + row) * k1 * N;
                glwe_encrypt(K->glwe_sk, k, N, glwe_std, r, nullptr, glwe);
                if (K->lwe_sk[i]) glwe[(size_t)row * N] += (uint64_t)1 << (64 - P.pbs_logB * (j + 1));
            }
    }
    // ---- the three LWE keyswitching keys ---------------------------------------------------------------------
    const uint64_t *s_glwe = helm_si_client_glwe_secret(pbs_key), *s_lwe = helm_si_client_lwe_secret(pbs_key);
    const std::vector<uint64_t> pbs_big(s_glwe, s_glwe + (size_t)S.k * S.N), pbs_small(s_lwe, s_lwe + S.n);
    make_ksk(K->glwe_sk, K->lwe_sk, P.ks_l, P.ks_logB, lwe_std, src, 0x7800000, K->ksk);
    make_ksk(pbs_big, K->glwe_sk, S.ks_l, S.ks_logB, glwe_std, src, 0x7900000, K->to_wop);
    make_ksk(K->glwe_sk, pbs_small, S.ks_l, S.ks_logB, s_lwe_std, src, 0x7A00000, K->to_pbs);
    // ---- private functional packing keyswitching keys: [k+1][k*N+1][pfks_l][(k+1) N] ---------------------------
    //      key r, input element t (key element s'_t: the big key's bit, -1 for the body), level j: GLWE encryption of
    //      P_r * f_r(s'_t * 2^(64 - logB (j+1))),  (P_r, f_r) = (S_r, x -> -x) for r < k, (1, identity) for r = k
    const size_t glwe_words = (size_t)k1 * N;
    K->pfpksk.assign((size_t)k1 * (kN + 1) * P.pfks_l * glwe_words, 0);
    for (int r = 0; r < k1; r++) {
        #pragma omp parallel for schedule(dynamic, 8)
        for (int t = 0; t <= kN; t++) {
            Rng rg = src.stream(0x7B00000 + (uint64_t)r * 0x100000 + (uint64_t)t);
            std::vector<uint64_t> msg((size_t)N);
            const uint64_t key_element = t < kN ? K->glwe_sk[(size_t)t] : ~0ull;
            for (int j = 0; j < P.pfks_l; j++) {
                const uint64_t scaled = ((uint64_t)1 << (64 - P.pfks_logB * (j + 1))) * key_element;
                if (r < k)
                    for (int u = 0; u < N; u++) msg[(size_t)u] = K->glwe_sk[(size_t)r * N + u] * (0ull - scaled);
                else {
                    std::fill(msg.begin(), msg.end(), 0ull);
                    msg[0] = scaled;
                }
                glwe_encrypt(K->glwe_sk, k, N, glwe_std, rg, msg.data(),
                             K->pfpksk.data() + (((size_t)r * (kN + 1) + t) * P.pfks_l + j) * glwe_words);
            }
        }
    }
    *out = K.release();
    return 0;
}

void helm_wop_client_key_free(helm_wop_client_key *key) { delete key; }

int helm_wop_client_key_part(const helm_wop_client_key *key, int which, const uint64_t **words, size_t *n_words)
{
    if (!key || !words || !n_words) return failw(HELM_ERR_INVALID, "null argument");
    const std::vector<uint64_t> *v = nullptr;
    switch (which) {
    case HELM_WOP_KEY_BSK: v = &key->bsk; break;
    case HELM_WOP_KEY_KSK: v = &key->ksk; break;
    case HELM_WOP_KEY_KSK_TO_WOPBS: v = &key->to_wop; break;
    case HELM_WOP_KEY_KSK_TO_PBS: v = &key->to_pbs; break;
    case HELM_WOP_KEY_PFPKSK: v = &key->pfpksk; break;
    case HELM_WOP_KEY_LWE_SECRET: v = &key->lwe_sk; break;
    case HELM_WOP_KEY_GLWE_SECRET: v = &key->glwe_sk; break;
    default: return failw(HELM_ERR_INVALID, "unknown key part");
    }
    *words = v->data();
    *n_words = v->size();
    return 0;
}

} // extern "C"
