//! Standard-domain server keys for the engine, from HELM's own tfhe client key.
//!
//! `tfhe::boolean::ServerKey` keeps its bootstrapping key in the Fourier domain, which cannot be
//! imported (the engine converts standard-domain GGSWs to its own NTT domain on the device).  The shim
//! therefore regenerates both server-side keys from the ClientKey's secret keys with tfhe's core_crypto
//! generators and hands the containers' words to the C ABI through the layout converters
//! (`helm_keys_*_from_tfhe`, helm_amd/csrc/host/key_import.cpp).
//!
//! NOT COMPILED in this repository's image (no rustc).  Every tfhe item named below is [RECALLED] from
//! tfhe-rs 0.4 and has to be checked against the crate:
//!   tfhe::boolean::ClientKey::{lwe_secret_key, glwe_secret_key, parameters}      (private fields in 0.4:
//!       reachable through `boolean::engine::BooleanEngine` internals or a serde round trip of the key)
//!   core_crypto::algorithms::{par_allocate_and_generate_new_lwe_bootstrap_key,
//!                             allocate_and_generate_new_lwe_keyswitch_key}
//!   core_crypto::entities::{LweBootstrapKeyOwned<u32>, LweKeyswitchKeyOwned<u32>} + `.as_ref()` -> &[u32]
use helm_hip_sys as sys;
use tfhe::boolean::prelude::*;
use tfhe::core_crypto::prelude::*;

pub struct StandardKeys {
    pub params: sys::helm_hip_params,
    pub bsk: Vec<u32>, // this ABI's [n][pbs_l][k+1][k+1][N]
    pub ksk: Vec<u32>, // this ABI's [k*N][ks_l][n+1], level 1 first
}

/// `tfhe::boolean::gen_keys()` parameters (reference src/bin/helm.rs:241) as the engine's struct.
pub fn engine_params(p: &BooleanParameters) -> sys::helm_hip_params {
    sys::helm_hip_params {
        torus_bits: 32,
        n: p.lwe_dimension.0 as i32,
        k: p.glwe_dimension.0 as i32,
        N: p.polynomial_size.0 as i32,
        pbs_l: p.pbs_level.0 as i32,
        pbs_logB: p.pbs_base_log.0 as i32,
        ks_l: p.ks_level.0 as i32,
        ks_logB: p.ks_base_log.0 as i32,
        pbs_order: 0,
        grouping_factor: 1,
    }
}

/// Regenerate the standard-domain keys under the client's secret keys.
pub fn standard_keys(
    params: &BooleanParameters,
    lwe_sk: &LweSecretKeyOwned<u32>,
    glwe_sk: &GlweSecretKeyOwned<u32>,
    generator: &mut EncryptionRandomGenerator<ActivatedRandomGenerator>,
) -> StandardKeys {
    // [RECALLED] argument order of the 0.4 generators
    let bsk: LweBootstrapKeyOwned<u32> = par_allocate_and_generate_new_lwe_bootstrap_key(
        lwe_sk, glwe_sk, params.pbs_base_log, params.pbs_level, params.glwe_modular_std_dev,
        CiphertextModulus::new_native(), generator);
    let big_lwe_sk = glwe_sk.clone().into_lwe_secret_key();
    let ksk: LweKeyswitchKeyOwned<u32> = allocate_and_generate_new_lwe_keyswitch_key(
        &big_lwe_sk, lwe_sk, params.ks_base_log, params.ks_level, params.lwe_modular_std_dev,
        CiphertextModulus::new_native(), generator);

    let p = engine_params(params);
    let (src_b, src_k): (&[u32], &[u32]) = (bsk.as_ref(), ksk.as_ref());
    let mut out = StandardKeys { params: p, bsk: vec![0; src_b.len()], ksk: vec![0; src_k.len()] };
    // container order -> ABI order (the KSK's levels are stored last-to-first in tfhe: reversed here)
    check_keys(unsafe { sys::helm_keys_bsk32_from_tfhe(&p, src_b.as_ptr(), out.bsk.as_mut_ptr(), src_b.len()) });
    check_keys(unsafe { sys::helm_keys_ksk32_from_tfhe(&p, src_k.as_ptr(), out.ksk.as_mut_ptr(), src_k.len()) });
    out
}

fn check_keys(rc: i32) {
    if rc != 0 {
        let m = unsafe { std::ffi::CStr::from_ptr(sys::helm_keys_last_error()) };
        panic!("key import: {}", m.to_string_lossy());
    }
}

// ---- 64-bit torus: LUT mode (shortint) and arithmetic mode (integer) -------------------------------------------------

pub struct StandardKeys64 {
    pub params: sys::helm_si_params,
    pub bsk: Vec<u64>, // this ABI's [n][pbs_l][k+1][k+1][N]; multi-bit sets: [n/g][2^g][pbs_l][k+1][k+1][N]
    pub ksk: Vec<u64>, // this ABI's [k*N][ks_l][n+1], level 1 first
}

/// `tfhe::shortint::ClassicPBSParameters` (helm.rs:301: PARAM_MESSAGE_1_CARRY_1_KS_PBS) as the engine's struct.
pub fn engine_params64(p: &tfhe::shortint::ClassicPBSParameters) -> sys::helm_si_params {
    sys::helm_si_params {
        n: p.lwe_dimension.0 as i32, k: p.glwe_dimension.0 as i32, N: p.polynomial_size.0 as i32,
        pbs_l: p.pbs_level.0 as i32, pbs_logB: p.pbs_base_log.0 as i32,
        ks_l: p.ks_level.0 as i32, ks_logB: p.ks_base_log.0 as i32,
        message_modulus: p.message_modulus.0 as i32, carry_modulus: p.carry_modulus.0 as i32,
        grouping_factor: 1,
    }
}

/// Regenerate the standard-domain shortint keys under the client's secret keys (the server key holds its bootstrapping
/// key in the Fourier domain).  [RECALLED] generators as in `standard_keys`, with `u64` containers.  For the multi-bit
/// set of helm.rs:83 the bootstrapping key has to be generated in this ABI's subset-indicator convention
/// (include/helm_shortint.h): tfhe's own multi-bit key layout differs and is NOT importable word for word.
pub fn standard_keys64(
    params: &tfhe::shortint::ClassicPBSParameters,
    lwe_sk: &LweSecretKeyOwned<u64>,
    glwe_sk: &GlweSecretKeyOwned<u64>,
    generator: &mut EncryptionRandomGenerator<ActivatedRandomGenerator>,
) -> StandardKeys64 {
    let bsk: LweBootstrapKeyOwned<u64> = par_allocate_and_generate_new_lwe_bootstrap_key(
        lwe_sk, glwe_sk, params.pbs_base_log, params.pbs_level, params.glwe_modular_std_dev,
        CiphertextModulus::new_native(), generator);
    let big_lwe_sk = glwe_sk.clone().into_lwe_secret_key();
    let ksk: LweKeyswitchKeyOwned<u64> = allocate_and_generate_new_lwe_keyswitch_key(
        &big_lwe_sk, lwe_sk, params.ks_base_log, params.ks_level, params.lwe_modular_std_dev,
        CiphertextModulus::new_native(), generator);
    let p = engine_params64(params);
    let (src_b, src_k): (&[u64], &[u64]) = (bsk.as_ref(), ksk.as_ref());
    let mut out = StandardKeys64 { params: p, bsk: vec![0; src_b.len()], ksk: vec![0; src_k.len()] };
    check_keys(unsafe { sys::helm_keys_bsk64_from_tfhe(&p, src_b.as_ptr(), out.bsk.as_mut_ptr(), src_b.len()) });
    check_keys(unsafe { sys::helm_keys_ksk64_from_tfhe(&p, src_k.as_ptr(), out.ksk.as_mut_ptr(), src_k.len()) });
    out
}

/// MANDATORY on the shim's first build: the container orders above are recalled, not checked - a wrong
/// recollection imports keys that decrypt to garbage while every in-repository test stays green.  Encrypt a known value
/// with tfhe, run one bootstrap on the GPU with the imported keys, decrypt with tfhe; panic on mismatch.
pub fn import_self_check(client_key: &tfhe::shortint::ClientKey, keys: &StandardKeys64, device_id: i32) {
    let row_words = (keys.params.k * keys.params.N) as usize + 1;
    let t = (keys.params.message_modulus * keys.params.carry_modulus) as u64;
    let (mut ctx, mut wires) = (std::ptr::null_mut(), std::ptr::null_mut());
    unsafe {
        assert_eq!(sys::helm_si_ctx_create(device_id, &keys.params, &mut ctx), 0);
        assert_eq!(sys::helm_si_load_bootstrap_key(ctx, keys.bsk.as_ptr(), keys.bsk.len()), 0);
        assert_eq!(sys::helm_si_load_keyswitch_key(ctx, keys.ksk.as_ptr(), keys.ksk.len()), 0);
        assert_eq!(sys::helm_si_wires_alloc(ctx, 2 * t as i64, &mut wires), 0);
        for v in 0..t {
            let ct = client_key.encrypt(v);
            let (row, words): (i32, &[u64]) = (v as i32, ct.ct.as_ref());
            assert_eq!(sys::helm_si_wires_upload(ctx, wires, &row, words.as_ptr(), 1), 0);
        }
        // one 2-input gate per value pair would do as well; the identity of gates::lut()'s arity-1 path has no bootstrap,
        // so use arity 2 on (v, v): f(x, y) = table[(x & 1) * 2 + (y & 1)] with table 0x6 = XOR -> 0 for every v
        let arity = vec![2i32; t as usize];
        let in_idx: Vec<i32> = (0..t as i32).flat_map(|v| [v, v]).collect();
        let table = vec![0x6u64; t as usize];
        let out: Vec<i32> = (t as i32..2 * t as i32).collect();
        assert_eq!(sys::helm_si_eval_lut_level(ctx, wires, arity.as_ptr(), in_idx.as_ptr(), 2, table.as_ptr(), out.as_ptr(), t as i64), 0);
        assert_eq!(sys::helm_si_sync(ctx), 0);
        let mut words = vec![0u64; t as usize * row_words];
        assert_eq!(sys::helm_si_wires_download(ctx, wires, out.as_ptr(), words.as_mut_ptr(), t as i64), 0);
        for (v, w) in words.chunks(row_words).enumerate() {
            let got = client_key.decrypt(&shortint_from_words(client_key, w));
            assert_eq!(got, 0, "key import self-check: value {} came back as {} - the recalled container order is wrong", v, got);
        }
        sys::helm_si_wires_free(ctx, wires);
        sys::helm_si_ctx_destroy(ctx);
    }
}

/// [RECALLED] a shortint ciphertext around downloaded LWE words (degree = message_modulus - 1 after a bootstrap).
pub fn shortint_from_words(client_key: &tfhe::shortint::ClientKey, words: &[u64]) -> tfhe::shortint::Ciphertext {
    let lwe = LweCiphertextOwned::from_container(words.to_vec(), CiphertextModulus::new_native());
    let p = client_key.parameters;
    tfhe::shortint::Ciphertext::new(lwe, tfhe::shortint::ciphertext::Degree(p.message_modulus().0 - 1),
                                    p.message_modulus(), p.carry_modulus())
}

/// [RECALLED] the blocks of `FheUintN::try_encrypt(v)`: block i encrypts (v >> 2 i) & 3 (least significant first).
pub fn radix_block_words(client_key: &tfhe::ClientKey, v: u128, blocks: usize) -> Vec<u64> {
    let ck = client_key.integer_key().expect("integers enabled (helm.rs:81-86)"); // [RECALLED] accessor
    let radix = ck.encrypt_radix(v, blocks);
    radix.blocks().iter().flat_map(|b| b.ct.as_ref().to_vec()).collect()
}

/// Inverse of `radix_block_words` on downloaded rows.
pub fn radix_decrypt(client_key: &tfhe::ClientKey, words: &[u64], row_words: usize) -> u128 {
    let ck = client_key.integer_key().expect("integers enabled");
    let sck = ck.as_ref(); // [RECALLED] the shortint client key under the integer key
    words.chunks(row_words).enumerate().fold(0u128, |acc, (i, w)| {
        acc | ((sck.decrypt(&shortint_from_words(sck, w)) as u128 & 3) << (2 * i))
    })
}
