//! `plonky2` as the 0xPARC/plonky2-aes gadget crates see it, backed by the MI355X-native
//! prover `libp2aes.so` through the C ABI of `include/p2aes.h`.
//!
//! Only what the reference calls is here -- `CircuitBuilder`, `Target`/`BoolTarget`, `PartialWitness` + `WitnessWrite`,
//! `CircuitConfig`, `CircuitData::{prove, verify}`, `GoldilocksField` with the `Field`/`Field64`/`Sample` methods used, the
//! `PoseidonHash` sponge -- under the module paths the reference imports them from.  Each item cites the call site it serves.
//! The module `pod2` at the bottom re-exports the handful of pod2 names the ecgfp5 / poseidon-cipher crates use.
//!
//! This file has not been compiled in this repository's image (no Rust toolchain); `ffi.rs` is generated from the header.
#![allow(clippy::new_without_default)]

pub mod ffi;

use std::ffi::CStr;
use std::marker::PhantomData;

fn last_error() -> anyhow::Error {
    let msg = unsafe { CStr::from_ptr(ffi::p2_last_error()) }.to_string_lossy().into_owned();
    anyhow::anyhow!(msg)
}

pub mod field {
    pub mod types {
        /// `plonky2::field::types::Field` -- the methods the reference calls (circuit_aes.rs:186,249; circuit_gcm.rs:339).
        pub trait Field: Copy + Eq + core::fmt::Debug {
            const ZERO: Self;
            const ONE: Self;
            const NEG_ONE: Self;
            fn from_canonical_u64(v: u64) -> Self;
            fn from_canonical_u8(v: u8) -> Self { Self::from_canonical_u64(v as u64) }
            fn to_canonical_u64(&self) -> u64;
        }
        /// `Field64::ORDER` (ecgfp5/src/lib.rs:56).
        pub trait Field64: Field {
            const ORDER: u64;
        }
        /// `Sample::rand` (poseidon-cipher/src/lib.rs:143-144): uniform field element from the OS RNG.
        pub trait Sample: Sized {
            fn rand() -> Self;
        }
    }
    pub mod goldilocks_field {
        use super::types::{Field, Field64, Sample};
        /// Canonical representative in [0, p), p = 2^64 - 2^32 + 1.
        #[derive(Copy, Clone, Debug, Default, Eq, PartialEq, Hash)]
        pub struct GoldilocksField(pub u64);
        impl Field for GoldilocksField {
            const ZERO: Self = GoldilocksField(0);
            const ONE: Self = GoldilocksField(1);
            const NEG_ONE: Self = GoldilocksField(<Self as Field64>::ORDER - 1);
            fn from_canonical_u64(v: u64) -> Self { debug_assert!(v < <Self as Field64>::ORDER); GoldilocksField(v) }
            fn to_canonical_u64(&self) -> u64 { self.0 }
        }
        impl Field64 for GoldilocksField {
            const ORDER: u64 = 0xFFFF_FFFF_0000_0001;
        }
        impl Sample for GoldilocksField {
            fn rand() -> Self {
                use rand::RngCore;
                loop {
                    let v = rand::rngs::OsRng.next_u64();
                    if v < <Self as Field64>::ORDER { return GoldilocksField(v); }
                }
            }
        }
    }
    pub mod extension {
        /// Marker only: the backend fixes D = 2 (`standard_recursion_config`), the gadgets never touch extension elements.
        pub trait Extendable<const D: usize> {}
        impl Extendable<2> for super::goldilocks_field::GoldilocksField {}
        pub mod quintic {
            /// GF(p^5) element as its five coefficients (ecgfp5/src/lib.rs:64-69: `from_basefield_array`).
            #[derive(Copy, Clone, Debug, Eq, PartialEq)]
            pub struct QuinticExtension<F>(pub [F; 5]);
            impl<F: Copy> QuinticExtension<F> {
                pub fn from_basefield_array(a: [F; 5]) -> Self { QuinticExtension(a) }
            }
        }
    }
}

pub mod hash {
    pub mod hash_types {
        pub trait RichField: crate::field::types::Field64 {}
        impl RichField for crate::field::goldilocks_field::GoldilocksField {}
    }
    pub mod poseidon {
        /// Marker for `builder.hash_n_to_m_no_pad::<PoseidonHash>` (poseidon-cipher/src/circuit.rs:123).
        pub struct PoseidonHash;
        pub struct PoseidonPermutation<F>(core::marker::PhantomData<F>);
    }
    pub mod hashing {
        use crate::field::goldilocks_field::GoldilocksField as F;
        /// Native `hash_n_to_m_no_pad::<F, PoseidonPermutation<_>>` (poseidon-cipher/src/lib.rs:116).
        pub fn hash_n_to_m_no_pad(inputs: &[F], m: usize) -> Vec<F> {
            let inp: Vec<u64> = inputs.iter().map(|x| x.0).collect();
            let mut out = vec![0u64; m];
            unsafe { crate::ffi::p2_native_hash_n_to_m_no_pad(inp.as_ptr(), inp.len(), out.as_mut_ptr(), m) };
            out.into_iter().map(F).collect()
        }
    }
}

pub mod iop {
    pub mod target {
        /// Opaque handle of the backend: a virtual target index, or bit 63 | row << 8 | column for a routed wire.
        #[derive(Copy, Clone, Debug, Eq, PartialEq, Hash)]
        pub struct Target(pub u64);
        #[derive(Copy, Clone, Debug, Eq, PartialEq, Hash)]
        pub struct BoolTarget { pub target: Target }
        impl BoolTarget {
            pub fn new_unsafe(target: Target) -> Self { BoolTarget { target } }   // circuit_gcm.rs:424
        }
    }
    pub mod witness {
        use super::target::Target;
        use crate::field::types::Field;
        use core::marker::PhantomData;
        /// `PartialWitness::new` + `set_target` (circuit_aes.rs:283,401): the sparse input map handed to `prove`.
        pub struct PartialWitness<F> {
            pub(crate) targets: Vec<u64>,
            pub(crate) values: Vec<u64>,
            _f: PhantomData<F>,
        }
        impl<F> PartialWitness<F> {
            pub fn new() -> Self { PartialWitness { targets: Vec::new(), values: Vec::new(), _f: PhantomData } }
        }
        pub trait WitnessWrite<F: Field> {
            fn set_target(&mut self, target: Target, value: F) -> anyhow::Result<()>;
            fn set_target_arr(&mut self, targets: &[Target], values: &[F]) -> anyhow::Result<()> {
                anyhow::ensure!(targets.len() == values.len(), "set_target_arr: length mismatch");
                for (t, v) in targets.iter().zip(values) { self.set_target(*t, *v)?; }
                Ok(())
            }
        }
        impl<F: Field> WitnessWrite<F> for PartialWitness<F> {
            /// Setting a target twice is fine when the values agree and an error otherwise, as in plonky2.
            fn set_target(&mut self, target: Target, value: F) -> anyhow::Result<()> {
                let v = value.to_canonical_u64();
                if let Some(i) = self.targets.iter().position(|t| *t == target.0) {
                    anyhow::ensure!(self.values[i] == v, "target {:?} set twice with different values", target);
                    return Ok(());
                }
                self.targets.push(target.0);
                self.values.push(v);
                Ok(())
            }
        }
    }
}

pub mod plonk {
    pub mod config {
        /// `PoseidonGoldilocksConfig` -- the only configuration the backend implements.
        pub struct PoseidonGoldilocksConfig;
    }
    pub mod circuit_data {
        use crate::field::types::Field;
        use crate::iop::witness::PartialWitness;
        use crate::{ffi, last_error};
        use core::marker::PhantomData;

        /// `CircuitConfig::standard_recursion_config()` / `standard_recursion_zk_config()` (circuit_gcm.rs:757,
        /// examples/aes_gcm_128.rs:36) -- the two configurations the reference instantiates.
        #[derive(Clone, Copy, Debug)]
        pub struct CircuitConfig { pub zero_knowledge: bool }
        impl CircuitConfig {
            pub fn standard_recursion_config() -> Self { CircuitConfig { zero_knowledge: false } }
            pub fn standard_recursion_zk_config() -> Self { CircuitConfig { zero_knowledge: true } }
        }

        /// Serialised proof exactly as the device writes it ("Proof layout").
        pub struct ProofWithPublicInputs<F, C, const D: usize> {
            pub bytes: Vec<u8>,
            pub(crate) _p: PhantomData<(F, C)>,
        }
        impl<F, C, const D: usize> ProofWithPublicInputs<F, C, D> {
            pub fn to_bytes(&self) -> Vec<u8> { self.bytes.clone() }
        }

        /// Result of `builder.build::<PoseidonGoldilocksConfig>()`: the compiled circuit, resident on one MI355X.
        pub struct CircuitData<F, C, const D: usize> {
            pub(crate) handle: *mut ffi::P2Circuit,
            pub(crate) blob: Vec<u8>,
            pub(crate) verifier_data: Vec<u64>,
            pub(crate) _p: PhantomData<(F, C)>,
        }
        // `prove(&self)` takes a shared reference and the handle serialises its own enqueues (include/p2aes.h).
        unsafe impl<F, C, const D: usize> Send for CircuitData<F, C, D> {}
        unsafe impl<F, C, const D: usize> Sync for CircuitData<F, C, D> {}
        impl<F, C, const D: usize> Drop for CircuitData<F, C, D> {
            fn drop(&mut self) { unsafe { ffi::p2_circuit_free(self.handle) } }
        }
        impl<F: Field, C, const D: usize> CircuitData<F, C, D> {
            /// circuit_gcm.rs:781 and the 19 other `.prove(` sites.  `Err` when witness generation
            /// fails -- the behaviour circuit_aes.rs:403-405 pins.
            pub fn prove(&self, pw: PartialWitness<F>) -> anyhow::Result<ProofWithPublicInputs<F, C, D>> {
                let mut proofs = self.prove_batch(&[pw])?;
                proofs.pop().unwrap()
            }
            /// Many witnesses of one circuit in one call -- the shape of ecgfp5/src/elgamal/circuit.rs:76-96 -- which is
            /// what fills the GPU.  One `Result` per witness.
            pub fn prove_batch(&self, pws: &[PartialWitness<F>]) -> anyhow::Result<Vec<anyhow::Result<ProofWithPublicInputs<F, C, D>>>> {
                let pb = unsafe { ffi::p2_circuit_proof_bytes(self.handle) };
                let asg: Vec<ffi::P2Assignment> = pws.iter()
                    .map(|pw| ffi::P2Assignment { targets: pw.targets.as_ptr(), values: pw.values.as_ptr(), count: pw.targets.len() })
                    .collect();
                let mut bytes = vec![0u8; pb * pws.len()];
                let mut status = vec![0 as std::os::raw::c_int; pws.len()];
                let rc = unsafe { ffi::p2_prove_batch(self.handle, pws.len(), asg.as_ptr(), bytes.as_mut_ptr(), status.as_mut_ptr()) };
                if rc != ffi::P2_OK { return Err(last_error()); }
                Ok(status.iter().enumerate().map(|(i, st)| match *st {
                    0 => Ok(ProofWithPublicInputs { bytes: bytes[i * pb..(i + 1) * pb].to_vec(), _p: PhantomData }),
                    1 => Err(anyhow::anyhow!("witness generation failed: conflicting value or lookup input not in table")),
                    2 => Err(anyhow::anyhow!("witness generation failed: a generator never ran (an input target was not set)")),
                    3 => Err(anyhow::anyhow!("Opening point is in the subgroup.")),
                    s => Err(anyhow::anyhow!("prove failed with status {s}")),
                }).collect())
            }
            /// circuit_gcm.rs:782 (19 sites).
            pub fn verify(&self, proof: ProofWithPublicInputs<F, C, D>) -> anyhow::Result<()> {
                let rc = unsafe {
                    ffi::p2_verify(self.blob.as_ptr(), self.blob.len(), self.verifier_data.as_ptr(), self.verifier_data.len(),
                                   proof.bytes.as_ptr(), proof.bytes.len())
                };
                if rc != ffi::P2_OK { return Err(last_error()); }
                Ok(())
            }
        }
    }
    pub mod circuit_builder {
        use super::circuit_data::{CircuitConfig, CircuitData};
        use crate::field::types::Field;
        use crate::iop::target::{BoolTarget, Target};
        use crate::{ffi, last_error};
        use core::marker::PhantomData;
        use std::sync::Arc;

        /// `CircuitBuilder::<F, D>::new(config)` and the methods of SURVEY.md A.1, one FFI call each.
        pub struct CircuitBuilder<F, const D: usize> { pub(crate) h: *mut ffi::P2Builder, _f: PhantomData<F> }
        impl<F, const D: usize> Drop for CircuitBuilder<F, D> {
            fn drop(&mut self) { unsafe { ffi::p2_builder_free(self.h) } }
        }
        impl<F: Field, const D: usize> CircuitBuilder<F, D> {
            pub fn new(config: CircuitConfig) -> Self {
                let h = unsafe { if config.zero_knowledge { ffi::p2_builder_new_zk() } else { ffi::p2_builder_new() } };
                CircuitBuilder { h, _f: PhantomData }
            }
            pub fn add_virtual_target(&mut self) -> Target { Target(unsafe { ffi::p2_builder_add_virtual_target(self.h) }) }   // circuit_aes.rs:178
            pub fn add_virtual_target_arr<const N: usize>(&mut self) -> [Target; N] { core::array::from_fn(|_| self.add_virtual_target()) }
            pub fn constant(&mut self, c: F) -> Target { Target(unsafe { ffi::p2_builder_constant(self.h, c.to_canonical_u64()) }) }   // :186
            pub fn zero(&mut self) -> Target { Target(unsafe { ffi::p2_builder_zero(self.h) }) }
            pub fn one(&mut self) -> Target { Target(unsafe { ffi::p2_builder_one(self.h) }) }
            /// c * x + y (circuit_aes.rs:249,356; circuit_gcm.rs:339,342,423)
            pub fn mul_const_add(&mut self, c: F, x: Target, y: Target) -> Target {
                Target(unsafe { ffi::p2_builder_mul_const_add(self.h, c.to_canonical_u64(), x.0, y.0) })
            }
            pub fn add(&mut self, x: Target, y: Target) -> Target { Target(unsafe { ffi::p2_builder_add(self.h, x.0, y.0) }) }   // circuit_gcm.rs:361
            pub fn sub(&mut self, x: Target, y: Target) -> Target { Target(unsafe { ffi::p2_builder_sub(self.h, x.0, y.0) }) }
            pub fn mul(&mut self, x: Target, y: Target) -> Target { Target(unsafe { ffi::p2_builder_mul(self.h, x.0, y.0) }) }   // :363
            pub fn is_equal(&mut self, x: Target, y: Target) -> BoolTarget {                                                      // :362
                BoolTarget::new_unsafe(Target(unsafe { ffi::p2_builder_is_equal(self.h, x.0, y.0) }))
            }
            pub fn select(&mut self, b: BoolTarget, x: Target, y: Target) -> Target {                                             // :314,323,364
                Target(unsafe { ffi::p2_builder_select(self.h, b.target.0, x.0, y.0) })
            }
            pub fn connect(&mut self, x: Target, y: Target) { unsafe { ffi::p2_builder_connect(self.h, x.0, y.0) } }              // :163
            /// circuit_aes.rs:300,317,334; circuit_gcm.rs:396,415.  `(u16, u16)` is two adjacent u16 in memory.
            pub fn add_lookup_table_from_pairs(&mut self, table: Arc<Vec<(u16, u16)>>) -> usize {
                // (u16, u16) has no guaranteed layout (repr(Rust)): flatten to the [in, out, in, out, ...] array the C ABI reads
                let flat: Vec<u16> = table.iter().flat_map(|&(a, b)| [a, b]).collect();
                unsafe { ffi::p2_builder_add_lookup_table_from_pairs(self.h, flat.as_ptr(), table.len()) }
            }
            /// circuit_aes.rs:182,193,250,357; circuit_gcm.rs:403,424.
            pub fn add_lookup_from_index(&mut self, looking_in: Target, lut_index: usize) -> Target {
                let t = unsafe { ffi::p2_builder_add_lookup_from_index(self.h, looking_in.0, lut_index) };
                assert!(t != u64::MAX, "{}", last_error());
                Target(t)
            }
            /// `hash_n_to_m_no_pad::<PoseidonHash>` (poseidon-cipher/src/circuit.rs:123, hashed_elgamal/circuit.rs:42).
            pub fn hash_n_to_m_no_pad<H>(&mut self, inputs: Vec<Target>, m: usize) -> Vec<Target> {
                let inp: Vec<u64> = inputs.iter().map(|t| t.0).collect();
                let mut out = vec![0u64; m];
                let rc = unsafe { ffi::p2_builder_hash_n_to_m_no_pad(self.h, inp.as_ptr(), inp.len(), out.as_mut_ptr(), m) };
                assert!(rc == ffi::P2_OK, "{}", last_error());
                out.into_iter().map(Target).collect()
            }
            pub fn num_gates(&self) -> usize { unsafe { ffi::p2_builder_num_gates(self.h) } }
            /// `build::<PoseidonGoldilocksConfig>()` (19 sites): compile on the host, then upload the circuit
            /// to HIP device P2AES_DEVICE (default 0) and commit constants | sigmas there.
            pub fn build<C>(self) -> CircuitData<F, C, D> {
                let (mut blob_ptr, mut len) = (core::ptr::null_mut::<u8>(), 0usize);
                let rc = unsafe { ffi::p2_builder_build(self.h, &mut blob_ptr, &mut len) };
                assert!(rc == ffi::P2_OK, "{}", last_error());
                let blob = unsafe { std::slice::from_raw_parts(blob_ptr, len) }.to_vec();
                unsafe { ffi::p2_blob_free(blob_ptr) };
                let device = std::env::var("P2AES_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
                let handle = unsafe { ffi::p2_circuit_load(blob.as_ptr(), blob.len(), device) };
                assert!(!handle.is_null(), "{}", last_error());   // no HIP device: the prover has no CPU fallback
                let mut vd = vec![0u64; 80];
                let mut n = 0usize;
                let rc = unsafe { ffi::p2_circuit_verifier_data(handle, vd.as_mut_ptr(), vd.len(), &mut n) };
                assert!(rc == ffi::P2_OK, "{}", last_error());
                vd.truncate(n);
                CircuitData { handle, blob, verifier_data: vd, _p: PhantomData }
            }
        }
    }
}

/// The pod2 names the ecgfp5 and poseidon-cipher crates import (`pod2::backends::plonky2::primitives::ec::{curve, bits}`,
/// SURVEY.md A.1 last rows), over `p2_ecgfp5_*` and `p2_builder_*point*`.  Workspace line: `pod2 = { path = ..., package = "plonky2-hip" }`
/// with `pub use plonky2::pod2 as backends` style re-exports in a two-line facade crate.
pub mod pod2 {
    pub mod curve {
        use crate::ffi;
        use crate::iop::target::{BoolTarget, Target};
        use crate::plonk::circuit_builder::CircuitBuilder;
        use crate::field::types::Field;

        /// Affine (x, u) coordinates over GF(p^5), ten words (`Point::as_fields` = x ++ u, hashed_elgamal.rs:23).
        #[derive(Copy, Clone, Debug, Eq, PartialEq)]
        pub struct Point { pub x: [u64; 5], pub u: [u64; 5] }
        fn pack(p: &Point) -> [u64; 10] { let mut o = [0u64; 10]; o[..5].copy_from_slice(&p.x); o[5..].copy_from_slice(&p.u); o }
        fn unpack(o: [u64; 10]) -> Point { let mut p = Point { x: [0; 5], u: [0; 5] }; p.x.copy_from_slice(&o[..5]); p.u.copy_from_slice(&o[5..]); p }
        /// GROUP_ORDER as five little-endian 64-bit limbs (ecgfp5/src/lib.rs:12).
        pub fn group_order() -> [u64; 5] { let mut o = [0u64; 5]; unsafe { ffi::p2_ecgfp5_group_order(o.as_mut_ptr()) }; o }
        impl Point {
            pub fn generator() -> Self { let mut o = [0u64; 10]; unsafe { ffi::p2_ecgfp5_generator(o.as_mut_ptr()) }; unpack(o) }
            pub fn new_rand_from_subgroup() -> Self {
                let mut o = [0u64; 10];
                // OS randomness can fail: an all-zero buffer must never pass for a random point or key
                let rc = unsafe { ffi::p2_ecgfp5_random_point(o.as_mut_ptr()) };
                assert!(rc == ffi::P2_OK, "{}", crate::last_error());
                unpack(o)
            }
            pub fn as_fields(&self) -> Vec<u64> { pack(self).to_vec() }
            pub fn inverse(&self) -> Self { let mut o = [0u64; 10]; unsafe { ffi::p2_ecgfp5_neg(pack(self).as_ptr(), o.as_mut_ptr()) }; unpack(o) }   // elgamal.rs:21
            pub fn is_in_subgroup(&self) -> bool { unsafe { ffi::p2_ecgfp5_is_in_subgroup(pack(self).as_ptr()) == 1 } }
            pub fn compress_from_subgroup(&self) -> [u64; 5] { let mut w = [0u64; 5]; unsafe { ffi::p2_ecgfp5_compress(pack(self).as_ptr(), w.as_mut_ptr()) }; w }   // lib.rs:82
            pub fn decompress_into_subgroup(w: &[u64; 5]) -> anyhow::Result<Self> {                                                                                     // lib.rs:74
                let mut o = [0u64; 10];
                anyhow::ensure!(unsafe { ffi::p2_ecgfp5_decompress(w.as_ptr(), o.as_mut_ptr()) } == ffi::P2_OK, "not the encoding of a group element");
                Ok(unpack(o))
            }
            /// `&k * P` with k as five little-endian limbs (elgamal.rs:13-15).
            pub fn mul_scalar(&self, k: &[u64; 5]) -> Self { let mut o = [0u64; 10]; unsafe { ffi::p2_ecgfp5_mul(k.as_ptr(), pack(self).as_ptr(), o.as_mut_ptr()) }; unpack(o) }
        }
        impl core::ops::Add for Point {
            type Output = Point;
            fn add(self, q: Point) -> Point { let mut o = [0u64; 10]; unsafe { ffi::p2_ecgfp5_add(pack(&self).as_ptr(), pack(&q).as_ptr(), o.as_mut_ptr()) }; unpack(o) }
        }
        #[derive(Copy, Clone, Debug)]
        pub struct PointTarget(pub [Target; 10]);
        #[derive(Clone, Debug)]
        pub struct BigUInt320Target { pub bits: Vec<BoolTarget> }
        fn raw<const N: usize>(ts: &[Target]) -> [u64; N] { core::array::from_fn(|i| ts[i].0) }
        /// `CircuitBuilderElliptic` / `CircuitBuilderBits` (ecgfp5/src/circuit.rs:33-38, elgamal/circuit.rs:34-37,70-72).
        pub trait CircuitBuilderElliptic {
            fn add_virtual_point_target(&mut self) -> PointTarget;
            fn constant_point(&mut self, p: Point) -> PointTarget;
            fn add_virtual_biguint320_target(&mut self) -> BigUInt320Target;
            fn multiply_point(&mut self, bits: &BigUInt320Target, p: &PointTarget) -> PointTarget;
            fn add_point(&mut self, p: &PointTarget, q: &PointTarget) -> PointTarget;
        }
        impl<F: Field, const D: usize> CircuitBuilderElliptic for CircuitBuilder<F, D> {
            fn add_virtual_point_target(&mut self) -> PointTarget {
                let mut o = [0u64; 10];
                unsafe { ffi::p2_builder_add_virtual_point_target(self.h, o.as_mut_ptr()) };
                PointTarget(o.map(Target))
            }
            fn constant_point(&mut self, p: Point) -> PointTarget {
                let mut o = [0u64; 10];
                unsafe { ffi::p2_builder_constant_point(self.h, pack(&p).as_ptr(), o.as_mut_ptr()) };
                PointTarget(o.map(Target))
            }
            fn add_virtual_biguint320_target(&mut self) -> BigUInt320Target {
                let mut o = [0u64; 320];
                unsafe { ffi::p2_builder_add_virtual_biguint320_target(self.h, o.as_mut_ptr()) };
                BigUInt320Target { bits: o.iter().map(|t| BoolTarget::new_unsafe(Target(*t))).collect() }
            }
            fn multiply_point(&mut self, bits: &BigUInt320Target, p: &PointTarget) -> PointTarget {
                let b: Vec<u64> = bits.bits.iter().map(|t| t.target.0).collect();
                let mut o = [0u64; 10];
                let rc = unsafe { ffi::p2_builder_multiply_point(self.h, b.as_ptr(), raw::<10>(&p.0).as_ptr(), o.as_mut_ptr()) };
                assert!(rc == ffi::P2_OK, "{}", crate::last_error());
                PointTarget(o.map(Target))
            }
            fn add_point(&mut self, p: &PointTarget, q: &PointTarget) -> PointTarget {
                let mut o = [0u64; 10];
                unsafe { ffi::p2_builder_add_point(self.h, raw::<10>(&p.0).as_ptr(), raw::<10>(&q.0).as_ptr(), o.as_mut_ptr()) };
                PointTarget(o.map(Target))
            }
        }
    }
}

#[allow(unused)]
fn _phantom_use(_: PhantomData<()>) {}
