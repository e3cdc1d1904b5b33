/*
 * helm_client.h — client-side key material, encryption and decryption (CPU).
 *
 * In the reference these are the `tfhe` client/server key objects:
 *   tfhe::boolean::gen_keys()                      reference src/bin/helm.rs:241
 *   ClientKey::encrypt / ClientKey::decrypt        reference src/circuit.rs:463-476, 558
 *   concrete-core key generation (feature "gpu")   reference src/bin/helm.rs:154-186
 * Key generation and (en/de)cryption are client operations and run on the CPU in
 * the reference too; the server-side hot path is include/helm_hip.h.
 *
 * The "server key" is exported in the standard (coefficient) domain in the
 * layouts include/helm_hip.h documents, ready for helm_hip_load_*_key().
 * Randomness (helm_amd/csrc/rng.hpp): `seed` = HELM_SEED_OS_ENTROPY (0) draws a 256-bit key
 * from the operating system (getrandom) and runs ChaCha20 streams under it for the secret
 * keys, the key material's masks and noise; the client key's encryption generator gets its
 * own, independent OS key - as tfhe's gen_keys() seeds from the OS (helm.rs:241,301).
 * Any other seed selects a DETERMINISTIC, INSECURE xoshiro256** generator: for tests,
 * golden vectors and reproducible benchmarks only (tests/circuit_test.rs:119 fixes its
 * seed for the same reason).
 */
#ifndef HELM_CLIENT_H
#define HELM_CLIENT_H

#include <stddef.h>
#include <stdint.h>
#include "helm_hip.h"
#include "helm_shortint.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct helm_client_key helm_client_key;

#define HELM_SEED_OS_ENTROPY 0ull

/* 0 when the ChaCha20 block function reproduces the RFC 8439 section 2.3.2 vector. */
int helm_client_rng_selftest(void);

/* Named parameter sets.  "boolean_default": tfhe 0.4.1 boolean::DEFAULT_PARAMETERS
 * (what gen_keys() uses, helm.rs:241) as recalled in SURVEY.md App. B;
 * "helm_cuda": the set hard-coded at helm.rs:141-146.  Returns 0 or HELM_ERR_INVALID. */
int helm_client_named_params(const char *name, helm_hip_params *params, double *lwe_noise_std,
                             double *glwe_noise_std);

const char *helm_client_last_error(void);

/* Generate the LWE secret key (n bits), the GLWE secret key (k*N bits), the
 * bootstrapping key and the keyswitching key. Noise standard deviations are
 * relative to the torus (as in tfhe's StandardDev). */
int helm_client_keygen(const helm_hip_params *params, double lwe_noise_std, double glwe_noise_std,
                       uint64_t seed, helm_client_key **out);
void helm_client_key_free(helm_client_key *key);

/* The parameter set the key was generated for. */
int helm_client_params(const helm_client_key *key, helm_hip_params *out);
size_t helm_client_bsk_words(const helm_client_key *key);
size_t helm_client_ksk_words(const helm_client_key *key);
const uint32_t *helm_client_bsk(const helm_client_key *key); /* [n][pbs_l][k+1][k+1][N] */
const uint32_t *helm_client_ksk(const helm_client_key *key); /* [k*N][ks_l][n+1]       */
/* Secret key bits as 0/1 words (tests and the oracle's decrypt use them). */
const uint32_t *helm_client_lwe_secret(const helm_client_key *key);  /* n   */
const uint32_t *helm_client_glwe_secret(const helm_client_key *key); /* k*N */

/* ClientKey::encrypt(bool): rows of n+1 words under the small key. */
int helm_client_encrypt_bool(helm_client_key *key, const uint8_t *bits, int64_t count, uint32_t *lwe_out);
/* ClientKey::decrypt: phase < 2^31 => true (reference src/circuit.rs:948). */
int helm_client_decrypt_bool(const helm_client_key *key, const uint32_t *lwe, int64_t count, uint8_t *bits_out);
/* Raw phases b - <a,s> (noise measurements). big != 0: ciphertexts under the k*N key. */
int helm_client_phase(const helm_client_key *key, const uint32_t *lwe, int64_t count, int big, uint32_t *phase_out);

/* ---- shortint (LUT / arithmetic mode) client: 64-bit torus, KS_PBS order ------------
 * tfhe::shortint::gen_keys(PARAM_...)              reference src/bin/helm.rs:301
 * ClientKey::encrypt(u64) / decrypt                reference src/circuit.rs:982-996, 1092
 * Ciphertexts are encrypted under the BIG key (k*N words + body). */
typedef struct helm_si_client_key helm_si_client_key;
/* "shortint_m2c2": PARAM_MESSAGE_2_CARRY_2_KS_PBS [recalled; the 3-bit-capable
 * class tests/circuit_test.rs:287 needs]; "shortint_m2c2_multibit3":
 * PARAM_MULTI_BIT_MESSAGE_2_CARRY_2_GROUP_3_KS_PBS, the arithmetic-mode set of src/bin/helm.rs:83
 * [dimensions recalled]; "si_toy_*": small sets for exact oracle comparisons. */
int helm_si_client_named_params(const char *name, helm_si_params *params, double *lwe_noise_std,
                                double *glwe_noise_std);
int helm_si_client_keygen(const helm_si_params *params, double lwe_noise_std, double glwe_noise_std,
                          uint64_t seed, helm_si_client_key **out);
void helm_si_client_key_free(helm_si_client_key *key);
int helm_si_client_params(const helm_si_client_key *key, helm_si_params *out);
/* the noise standard deviations the key was generated with (fractions of the torus) */
int helm_si_client_noise(const helm_si_client_key *key, double *lwe_noise_std, double *glwe_noise_std);
size_t helm_si_client_bsk_words(const helm_si_client_key *key);
size_t helm_si_client_ksk_words(const helm_si_client_key *key);
const uint64_t *helm_si_client_bsk(const helm_si_client_key *key); /* [n][pbs_l][k+1][k+1][N]; multi-bit: [n/g][2^g][...] */
const uint64_t *helm_si_client_ksk(const helm_si_client_key *key); /* [k*N][ks_l][n+1]       */
const uint64_t *helm_si_client_lwe_secret(const helm_si_client_key *key);  /* n   words of 0/1 */
const uint64_t *helm_si_client_glwe_secret(const helm_si_client_key *key); /* k*N words of 0/1 */
/* value v (taken mod message_modulus*carry_modulus) -> big LWE with body += v * delta */
int helm_si_client_encrypt(helm_si_client_key *key, const uint64_t *values, int64_t count, uint64_t *lwe_out);
/* round(phase / delta) mod (message_modulus*carry_modulus): message AND carry;
 * ClientKey::decrypt is this value mod message_modulus */
int helm_si_client_decrypt(const helm_si_client_key *key, const uint64_t *lwe, int64_t count, uint64_t *values_out);
int helm_si_client_phase(const helm_si_client_key *key, const uint64_t *lwe, int64_t count, int small, uint64_t *phase_out);

/* ---- key import: tfhe-rs 0.4 containers <-> this ABI (helm_amd/csrc/host/key_import.cpp) ----------
 * What the Rust shim (rust/helm-hip) calls between tfhe's `LweBootstrapKey` / `LweKeyswitchKey`
 * containers (as_ref() words of the standard-domain keys generated from HELM's ClientKey,
 * reference src/bin/helm.rs:241,301) and helm_hip_load_*_key / helm_si_load_*_key.
 * [RECALLED] orders, stated in key_import.cpp and INTEGRATION.md: the BSK is the same order (copied);
 * the KSK stores each block's levels last-to-first, this ABI first-to-last (levels reversed).
 * n_words must be the key's size for the parameter set; 0 or HELM_ERR_INVALID (helm_keys_last_error()). */
const char *helm_keys_last_error(void);
int helm_keys_bsk32_from_tfhe(const helm_hip_params *p, const uint32_t *tfhe, uint32_t *abi, size_t n_words);
int helm_keys_bsk32_to_tfhe(const helm_hip_params *p, const uint32_t *abi, uint32_t *tfhe, size_t n_words);
int helm_keys_ksk32_from_tfhe(const helm_hip_params *p, const uint32_t *tfhe, uint32_t *abi, size_t n_words);
int helm_keys_ksk32_to_tfhe(const helm_hip_params *p, const uint32_t *abi, uint32_t *tfhe, size_t n_words);
int helm_keys_bsk64_from_tfhe(const helm_si_params *p, const uint64_t *tfhe, uint64_t *abi, size_t n_words);
int helm_keys_bsk64_to_tfhe(const helm_si_params *p, const uint64_t *abi, uint64_t *tfhe, size_t n_words);
int helm_keys_ksk64_from_tfhe(const helm_si_params *p, const uint64_t *tfhe, uint64_t *abi, size_t n_words);
int helm_keys_ksk64_to_tfhe(const helm_si_params *p, const uint64_t *abi, uint64_t *tfhe, size_t n_words);
/* Any 64-bit key stored as `blocks` x `levels` x rows of row_words words whose levels tfhe keeps last-to-first:
 * the keyswitching keys between two parameter sets and the private functional packing keyswitching keys of
 * include/helm_wopbs.h ([RECALLED] order).  Its own inverse. */
int helm_keys_levels64_reverse(size_t blocks, int32_t levels, size_t row_words, const uint64_t *src, uint64_t *dst,
                               size_t n_words);

#ifdef __cplusplus
}
#endif
#endif
