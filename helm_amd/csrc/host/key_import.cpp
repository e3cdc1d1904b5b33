// key_import.cpp — tfhe-rs 0.4 key containers <-> the layouts this ABI loads.
//
// The Rust shim (rust/helm-hip) takes HELM's own tfhe keys (Cargo.toml:18) and hands their words to
// helm_hip_load_*_key / helm_si_load_*_key.  The containers' element order is the contract a real
// integration depends on; tfhe's source is not in this image, so every statement about it is
// [RECALLED] (tfhe-rs 0.4 core_crypto) and must be confirmed against the crate when the shim is first
// built.  What this file makes testable today: the conversions are exact inverses, they are NOT the
// identity where the orders differ, and a key pushed through "tfhe order" and back bootstraps bit for
// bit like the original (tests/test_key_import.py).
//
//   LweBootstrapKey<Vec<u32|u64>>   [RECALLED] a GgswCiphertextList: input key bit i major; within a
//       GGSW the level matrices in order of DecompositionLevel 1..l (level 1 = weight 2^(w - beta) first;
//       the external product walks them in reverse "to match the decomposition iterator"); within a
//       level matrix k+1 rows, each a GLWE ciphertext = k mask polynomials then the body; row r < k
//       carries -S_r * m * 2^(w - beta j), row k carries m * 2^(w - beta j).
//       = this ABI's [n][pbs_l][k+1][k+1][N]: same order, copied as is.
//   LweKeyswitchKey<Vec<u32|u64>>   [RECALLED] input key bit t major; within a block the l_ks LWE
//       ciphertexts are stored from the LAST level to the first (generation zips the block with
//       `(1..=l).rev()`; the keyswitch zips the block with the decomposition iterator, which yields the
//       least significant level first); each ciphertext n mask words then the body.
//       This ABI: [k*N][ks_l][n+1] with level index 0 = level 1 -> the level order is reversed.
//   WoP-PBS keys (include/helm_wopbs.h)   [RECALLED] the two LweKeyswitchKeys between the parameter sets are
//       ordinary keyswitching keys (levels reversed, as above); LwePrivateFunctionalPackingKeyswitchKeyList:
//       key r major, then the kN+1 input elements (the body's element last), then the levels stored from the
//       LAST to the first as in a keyswitching key, each a GLWE ciphertext (k mask polynomials, body) ->
//       helm_keys_levels64_reverse(blocks = (k+1)(kN+1), levels = pfks_l, row = (k+1) N).
//   Fourier-domain keys (what boolean::ServerKey / shortint::ServerKey hold for the BSK) cannot be
//       imported: the shim regenerates the standard-domain key from the ClientKey's secret keys
//       (rust/helm-hip/src/keys.rs).
#include "../../../include/helm_client.h"

#include <cstring>
#include <string>

namespace {
thread_local std::string g_kerr;
int kfail(const char *m)
{
    g_kerr = m;
    return HELM_ERR_INVALID;
}

template <typename T> int bsk_copy(size_t want, const T *src, T *dst, size_t n_words)
{
    if (!src || !dst) return kfail("null argument");
    if (n_words != want) return kfail("bootstrapping key: wrong number of words for this parameter set");
    std::memmove(dst, src, n_words * sizeof(T));
    return 0;
}

template <typename T> int ksk_reverse_levels(int kN, int l, int n, const T *src, T *dst, size_t n_words)
{
    if (!src || !dst) return kfail("null argument");
    const size_t row = (size_t)n + 1;
    if (n_words != (size_t)kN * l * row) return kfail("keyswitching key: wrong number of words for this parameter set");
    if (src == dst) return kfail("keyswitching key conversion is not in place");
    for (int t = 0; t < kN; t++)
        for (int j = 0; j < l; j++)
            std::memcpy(dst + ((size_t)t * l + j) * row, src + ((size_t)t * l + (l - 1 - j)) * row, row * sizeof(T));
    return 0;
}
} // namespace

extern "C" {

const char *helm_keys_last_error(void) { return g_kerr.c_str(); }

int helm_keys_levels64_reverse(size_t blocks, int32_t levels, size_t row_words, const uint64_t *src, uint64_t *dst,
                               size_t n_words)
{
    if (!src || !dst) return kfail("null argument");
    if (levels < 1 || n_words != blocks * (size_t)levels * row_words) return kfail("key: wrong number of words for these dimensions");
    if (src == dst) return kfail("key conversion is not in place");
    for (size_t t = 0; t < blocks; t++)
        for (int j = 0; j < levels; j++)
            std::memcpy(dst + (t * levels + j) * row_words, src + (t * levels + (levels - 1 - j)) * row_words,
                        row_words * sizeof(uint64_t));
    return 0;
}

int helm_keys_bsk32_from_tfhe(const helm_hip_params *p, const uint32_t *tfhe, uint32_t *abi, size_t n_words)
{
    if (!p) return kfail("null argument");
    return bsk_copy<uint32_t>((size_t)p->n * p->pbs_l * (p->k + 1) * (p->k + 1) * p->N, tfhe, abi, n_words);
}
int helm_keys_bsk32_to_tfhe(const helm_hip_params *p, const uint32_t *abi, uint32_t *tfhe, size_t n_words)
{
    return helm_keys_bsk32_from_tfhe(p, abi, tfhe, n_words);
}
int helm_keys_ksk32_from_tfhe(const helm_hip_params *p, const uint32_t *tfhe, uint32_t *abi, size_t n_words)
{
    if (!p) return kfail("null argument");
    return ksk_reverse_levels<uint32_t>(p->k * p->N, p->ks_l, p->n, tfhe, abi, n_words);
}
int helm_keys_ksk32_to_tfhe(const helm_hip_params *p, const uint32_t *abi, uint32_t *tfhe, size_t n_words)
{
    return helm_keys_ksk32_from_tfhe(p, abi, tfhe, n_words); // reversing the levels is an involution
}

int helm_keys_bsk64_from_tfhe(const helm_si_params *p, const uint64_t *tfhe, uint64_t *abi, size_t n_words)
{
    if (!p) return kfail("null argument");
    if (p->grouping_factor > 1)
        return kfail("multi-bit bootstrapping keys: tfhe's per-group GGSW order and product convention are not "
                     "written down here (INTEGRATION.md); generate the key in this ABI's subset-indicator order");
    return bsk_copy<uint64_t>((size_t)p->n * p->pbs_l * (p->k + 1) * (p->k + 1) * p->N, tfhe, abi, n_words);
}
int helm_keys_bsk64_to_tfhe(const helm_si_params *p, const uint64_t *abi, uint64_t *tfhe, size_t n_words)
{
    return helm_keys_bsk64_from_tfhe(p, abi, tfhe, n_words);
}
int helm_keys_ksk64_from_tfhe(const helm_si_params *p, const uint64_t *tfhe, uint64_t *abi, size_t n_words)
{
    if (!p) return kfail("null argument");
    return ksk_reverse_levels<uint64_t>(p->k * p->N, p->ks_l, p->n, tfhe, abi, n_words);
}
int helm_keys_ksk64_to_tfhe(const helm_si_params *p, const uint64_t *abi, uint64_t *tfhe, size_t n_words)
{
    return helm_keys_ksk64_from_tfhe(p, abi, tfhe, n_words);
}

} // extern "C"
