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
