//! Rust face of `libsourmash_amd.so`: the `extern "C"` declarations of `include/sourmash.h` /
//! `include/sourmash_amd.h` that the hot path needs, and a `KmerMinHash` with the reference's
//! public fields, method names and signatures (reference `src/lib.rs:37-46, 141-513`).  EVERY
//! method forwards to a C symbol of the library -- the sketch state is handed over through the raw
//! push symbols, the call is made, and the state is read back -- so the numbers are the GPU
//! library's, never this crate's.  Each symbol bound here is also called from
//! `tests/c_abi_client.c`, which is what link-checks the forwards.
//!
//! SOURCE ONLY: the build image has no rustc/cargo, this crate has never been compiled there.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_void};

#[repr(C)]
pub struct RawKmerMinHash {
    _private: [u8; 0],
}

#[repr(C)]
pub struct RawIndex {
    _private: [u8; 0],
}

#[repr(C)]
pub struct SourmashStr {
    pub data: *mut c_char,
    pub len: usize,
    pub owned: bool,
}

extern "C" {
    // ---- include/sourmash.h (reference src/ffi.rs, src/utils.rs)
    pub fn kmerminhash_new(n: u32, k: u32, prot: bool, seed: u64, mx: u64, track_abundance: bool) -> *mut RawKmerMinHash;
    pub fn kmerminhash_free(ptr: *mut RawKmerMinHash);
    pub fn kmerminhash_add_hash(ptr: *mut RawKmerMinHash, h: u64);
    pub fn kmerminhash_merge(ptr: *mut RawKmerMinHash, other: *const RawKmerMinHash);
    pub fn kmerminhash_add_from(ptr: *mut RawKmerMinHash, other: *const RawKmerMinHash);
    pub fn kmerminhash_mins_push(ptr: *mut RawKmerMinHash, val: u64);
    pub fn kmerminhash_abunds_push(ptr: *mut RawKmerMinHash, val: u64);
    pub fn kmerminhash_get_mins(ptr: *mut RawKmerMinHash) -> *const u64;
    pub fn kmerminhash_get_mins_size(ptr: *mut RawKmerMinHash) -> usize;
    pub fn kmerminhash_get_abunds(ptr: *mut RawKmerMinHash) -> *const u64;
    pub fn kmerminhash_get_abunds_size(ptr: *mut RawKmerMinHash) -> usize;
    pub fn kmerminhash_track_abundance(ptr: *mut RawKmerMinHash) -> bool;
    pub fn kmerminhash_compare(ptr: *mut RawKmerMinHash, other: *const RawKmerMinHash) -> f64;
    pub fn kmerminhash_count_common(ptr: *mut RawKmerMinHash, other: *const RawKmerMinHash) -> u64;
    pub fn sourmash_init();
    pub fn sourmash_err_get_last_code() -> u32;
    pub fn sourmash_err_get_last_message() -> SourmashStr;
    pub fn sourmash_err_clear();
    pub fn sourmash_str_free(s: *mut SourmashStr);
    // ---- include/sourmash_amd.h (additive)
    pub fn smh_hash_words(bytes: *const c_char, offsets: *const u64, n: u32, seed: u64, out: *mut u64) -> i32;
    pub fn smh_check_compatible(ptr: *const RawKmerMinHash, other: *const RawKmerMinHash) -> i32;
    pub fn smh_add_sequence_len(ptr: *mut RawKmerMinHash, seq: *const c_char, len: u64, force: bool) -> i32;
    pub fn smh_add_sequences(ptr: *mut RawKmerMinHash, seq: *const c_char, offsets: *const u64, n_records: u32, force: bool) -> i32;
    pub fn smh_add_many(ptr: *mut RawKmerMinHash, hashes: *const u64, n: u64) -> i32;
    pub fn smh_add_many_with_abund(ptr: *mut RawKmerMinHash, hashes: *const u64, abunds: *const u64, n: u64) -> i32;
    pub fn smh_intersection(
        ptr: *const RawKmerMinHash, other: *const RawKmerMinHash, common_out: *mut *mut u64, n_common: *mut u64,
        union_size: *mut u64,
    ) -> i32;
    pub fn smh_compare_block(
        rows: *const *mut RawKmerMinHash, n_rows: u32, cols: *const *mut RawKmerMinHash, n_cols: u32,
        jaccard: *mut f64, common: *mut u64, size: *mut u64, count_common: *mut u64, containment: *mut f64,
    ) -> i32;
    pub fn smh_add_sequences_grouped(
        sketches: *const *mut RawKmerMinHash, n_sketches: u32, seq: *const c_char, offsets: *const u64,
        groups: *const u32, n_records: u32, force: bool,
    ) -> i32;
    pub fn smh_index_new(nodes: *const *mut RawKmerMinHash, n_nodes: u32) -> *mut RawIndex;
    pub fn smh_index_free(index: *mut RawIndex);
    pub fn smh_index_len(index: *const RawIndex) -> u32;
    pub fn smh_index_find(
        index: *mut RawIndex, query: *const RawKmerMinHash, threshold: f64, containment: bool,
        out_indices: *mut u32, out_count: *mut u32,
    ) -> i32;
}

extern "C" {
    fn free(p: *mut c_void);
}

/// Error of the library's thread-local slot (codes of reference `src/errors.rs:28-50`).  Stands where
/// the reference has `failure::Error`.
#[derive(Debug, Clone, PartialEq)]
pub struct Error {
    pub code: u32,
    pub message: String,
}
pub type SourmashError = Error;

fn take_error() -> Error {
    unsafe {
        let code = sourmash_err_get_last_code();
        let mut s = sourmash_err_get_last_message();
        let message = if s.data.is_null() {
            String::new()
        } else {
            String::from_utf8_lossy(std::slice::from_raw_parts(s.data as *const u8, s.len)).into_owned()
        };
        sourmash_str_free(&mut s);
        sourmash_err_clear();
        Error { code, message }
    }
}

fn check(rc: i32) -> Result<(), Error> {
    if rc != 0 {
        Err(take_error())
    } else {
        Ok(())
    }
}

/// Reference `_hash_murmur(kmer: &[u8], seed: u64) -> u64` (`src/lib.rs:33-35`); the bytes may hold NULs.
pub fn _hash_murmur(kmer: &[u8], seed: u64) -> u64 {
    let off = [0u64, kmer.len() as u64];
    let mut out = 0u64;
    let rc = unsafe { smh_hash_words(kmer.as_ptr() as *const c_char, off.as_ptr(), 1, seed, &mut out) };
    if rc != 0 {
        panic!("{}", take_error().message); // no device: the reference's function cannot fail, this one has no CPU path
    }
    out
}

/// Same public fields as the reference struct (`src/lib.rs:37-46`).
#[derive(Debug, Clone, PartialEq)]
pub struct KmerMinHash {
    pub num: u32,
    pub ksize: u32,
    pub is_protein: bool,
    pub seed: u64,
    pub max_hash: u64,
    pub mins: Vec<u64>,
    pub abunds: Option<Vec<u64>>,
}

impl Default for KmerMinHash {
    /// reference `src/lib.rs:48-60`
    fn default() -> KmerMinHash {
        KmerMinHash { num: 1000, ksize: 21, is_protein: false, seed: 42, max_hash: 0, mins: Vec::with_capacity(1000), abunds: None }
    }
}

struct Handle(*mut RawKmerMinHash);
impl Drop for Handle {
    fn drop(&mut self) {
        unsafe { kmerminhash_free(self.0) }
    }
}

impl KmerMinHash {
    /// reference `src/lib.rs:142-174`
    pub fn new(num: u32, ksize: u32, is_protein: bool, seed: u64, max_hash: u64, track_abundance: bool) -> KmerMinHash {
        let cap = if num > 0 { num as usize } else { 1000 };
        KmerMinHash {
            num, ksize, is_protein, seed, max_hash,
            mins: Vec::with_capacity(cap),
            abunds: if track_abundance { Some(Vec::with_capacity(cap)) } else { None },
        }
    }

    /// Library-side copy of the current state (raw pushes: no ordering check, like the reference ABI).
    fn to_handle(&self) -> Handle {
        unsafe {
            let h = kmerminhash_new(self.num, self.ksize, self.is_protein, self.seed, self.max_hash, self.abunds.is_some());
            for &m in &self.mins {
                kmerminhash_mins_push(h, m);
            }
            if let Some(ab) = &self.abunds {
                for &a in ab {
                    kmerminhash_abunds_push(h, a);
                }
            }
            Handle(h)
        }
    }

    fn read_back(&mut self, h: &Handle) {
        unsafe {
            let n = kmerminhash_get_mins_size(h.0);
            let p = kmerminhash_get_mins(h.0);
            self.mins = std::slice::from_raw_parts(p, n).to_vec();
            free(p as *mut c_void);
            if kmerminhash_track_abundance(h.0) {
                // (`merge` turns tracking on: quirk Q5, reference src/lib.rs:391-401)
                let na = kmerminhash_get_abunds_size(h.0);
                let pa = kmerminhash_get_abunds(h.0);
                self.abunds = Some(if pa.is_null() { Vec::new() } else { std::slice::from_raw_parts(pa, na).to_vec() });
                free(pa as *mut c_void);
            }
        }
    }

    /// Run `f` on a library copy of `self`, read the state back, surface the error slot.
    fn mutate<F: FnOnce(*mut RawKmerMinHash) -> i32>(&mut self, f: F) -> Result<(), Error> {
        let h = self.to_handle();
        unsafe { sourmash_err_clear() };
        let rc = f(h.0);
        self.read_back(&h);
        if rc != 0 || unsafe { sourmash_err_get_last_code() } != 0 {
            return Err(take_error());
        }
        Ok(())
    }

    /// reference `src/lib.rs:176-190`
    pub fn check_compatible(&self, other: &KmerMinHash) -> Result<bool, Error> {
        let (a, b) = (self.to_handle(), other.to_handle());
        check(unsafe { smh_check_compatible(a.0, b.0) }).map(|_| true)
    }

    /// reference `src/lib.rs:192-245`
    pub fn add_hash(&mut self, hash: u64) {
        let _ = self.mutate(|h| unsafe {
            kmerminhash_add_hash(h, hash);
            0
        });
    }

    /// reference `src/lib.rs:247-250`
    pub fn add_word(&mut self, word: &[u8]) {
        let hash = _hash_murmur(word, self.seed);
        self.add_hash(hash);
    }

    /// reference `src/lib.rs:252-305`: on `Err` the windows before the offending one have been added,
    /// exactly as there.
    pub fn add_sequence(&mut self, seq: &[u8], force: bool) -> Result<(), Error> {
        self.mutate(|h| unsafe { smh_add_sequence_len(h, seq.as_ptr() as *const c_char, seq.len() as u64, force) })
    }

    /// reference `src/lib.rs:307-403`
    pub fn merge(&mut self, other: &KmerMinHash) -> Result<(), Error> {
        let o = other.to_handle();
        self.mutate(|h| unsafe {
            kmerminhash_merge(h, o.0);
            0
        })
    }

    /// reference `src/lib.rs:405-410`
    pub fn add_from(&mut self, other: &KmerMinHash) -> Result<(), Error> {
        let o = other.to_handle();
        self.mutate(|h| unsafe {
            kmerminhash_add_from(h, o.0);
            0
        })
    }

    /// reference `src/lib.rs:412-417`
    pub fn add_many(&mut self, hashes: &[u64]) -> Result<(), Error> {
        self.mutate(|h| unsafe { smh_add_many(h, hashes.as_ptr(), hashes.len() as u64) })
    }

    /// reference `src/lib.rs:419-426`
    pub fn add_many_with_abund(&mut self, hashes: &[(u64, u64)]) -> Result<(), Error> {
        let hs: Vec<u64> = hashes.iter().map(|p| p.0).collect();
        let ab: Vec<u64> = hashes.iter().map(|p| p.1).collect();
        self.mutate(|h| unsafe { smh_add_many_with_abund(h, hs.as_ptr(), ab.as_ptr(), hs.len() as u64) })
    }

    /// reference `src/lib.rs:428-436`
    pub fn count_common(&self, other: &KmerMinHash) -> Result<u64, Error> {
        let (a, b) = (self.to_handle(), other.to_handle());
        unsafe {
            sourmash_err_clear();
            let c = kmerminhash_count_common(a.0, b.0);
            if sourmash_err_get_last_code() != 0 {
                return Err(take_error());
            }
            Ok(c)
        }
    }

    /// reference `src/lib.rs:438-468`: the common hashes and the size of the combined sketch.
    pub fn intersection(&self, other: &KmerMinHash) -> Result<(Vec<u64>, u64), Error> {
        let (a, b) = (self.to_handle(), other.to_handle());
        let mut p: *mut u64 = std::ptr::null_mut();
        let (mut n, mut size) = (0u64, 0u64);
        check(unsafe { smh_intersection(a.0, b.0, &mut p, &mut n, &mut size) })?;
        let common = unsafe { std::slice::from_raw_parts(p, n as usize).to_vec() };
        unsafe { free(p as *mut c_void) };
        Ok((common, size))
    }

    /// reference `src/lib.rs:470-499`
    pub fn intersection_size(&self, other: &KmerMinHash) -> Result<(u64, u64), Error> {
        let (a, b) = (self.to_handle(), other.to_handle());
        let (mut common, mut size) = (0u64, 0u64);
        let (rp, cp) = ([a.0], [b.0]);
        check(unsafe {
            smh_compare_block(rp.as_ptr(), 1, cp.as_ptr(), 1, std::ptr::null_mut(), &mut common, &mut size,
                              std::ptr::null_mut(), std::ptr::null_mut())
        })?;
        Ok((common, size))
    }

    /// reference `src/lib.rs:501-508`
    pub fn compare(&self, other: &KmerMinHash) -> Result<f64, Error> {
        let (a, b) = (self.to_handle(), other.to_handle());
        unsafe {
            sourmash_err_clear();
            let j = kmerminhash_compare(a.0, b.0);
            if sourmash_err_get_last_code() != 0 {
                return Err(take_error());
            }
            Ok(j)
        }
    }

    /// reference `src/lib.rs:510-512`
    pub fn size(&self) -> usize {
        let h = self.to_handle();
        unsafe { kmerminhash_get_mins_size(h.0) }
    }

    /// Many records in one device pass (no counterpart in the reference API).
    pub fn add_sequences(&mut self, records: &[&[u8]], force: bool) -> Result<(), Error> {
        let mut flat = Vec::new();
        let mut off = vec![0u64];
        for r in records {
            flat.extend_from_slice(r);
            off.push(flat.len() as u64);
        }
        self.mutate(|h| unsafe { smh_add_sequences(h, flat.as_ptr() as *const c_char, off.as_ptr(), records.len() as u32, force) })
    }
}

/// rows x cols Jaccard block in one launch (N^2 calls of `compare` in the reference).
pub fn compare_matrix(rows: &[KmerMinHash], cols: &[KmerMinHash]) -> Result<Vec<f64>, Error> {
    let rh: Vec<Handle> = rows.iter().map(|m| m.to_handle()).collect();
    let ch: Vec<Handle> = cols.iter().map(|m| m.to_handle()).collect();
    let rp: Vec<*mut RawKmerMinHash> = rh.iter().map(|h| h.0).collect();
    let cp: Vec<*mut RawKmerMinHash> = ch.iter().map(|h| h.0).collect();
    let mut out = vec![0f64; rows.len() * cols.len()];
    check(unsafe {
        smh_compare_block(rp.as_ptr(), rp.len() as u32, cp.as_ptr(), cp.len() as u32, out.as_mut_ptr(),
                          std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut(), std::ptr::null_mut())
    })?;
    Ok(out)
}

/// One signature per group of records (one genome = its contigs) in ONE device pass: the loop
/// `for file in files { let mut mh = template.clone(); for rec in file { mh.add_sequence(rec) } }`
/// of the reference's callers.  `groups[r]` is the sketch that record `r` feeds.
pub fn sketch_groups(sketches: &mut [KmerMinHash], records: &[&[u8]], groups: &[u32], force: bool) -> Result<(), Error> {
    assert_eq!(records.len(), groups.len());
    let mut flat = Vec::new();
    let mut off = vec![0u64];
    for r in records {
        flat.extend_from_slice(r);
        off.push(flat.len() as u64);
    }
    let hs: Vec<Handle> = sketches.iter().map(|m| m.to_handle()).collect();
    let hp: Vec<*mut RawKmerMinHash> = hs.iter().map(|h| h.0).collect();
    let rc = unsafe {
        smh_add_sequences_grouped(hp.as_ptr(), hp.len() as u32, flat.as_ptr() as *const c_char, off.as_ptr(),
                                  groups.as_ptr(), records.len() as u32, force)
    };
    for (m, h) in sketches.iter_mut().zip(hs.iter()) {
        m.read_back(h);
    }
    check(rc)
}

/// `LinearIndex` (reference `src/index/linear.rs`) whose leaves live in HBM: `find` uploads only
/// the query.  Built once from the leaves' sketches; `search_fn` of the reference becomes the
/// `containment` flag (similarity vs containment, `src/index/search.rs`).
pub struct ResidentIndex {
    raw: *mut RawIndex,
    _leaves: Vec<Handle>,
}

impl ResidentIndex {
    pub fn new(leaves: &[KmerMinHash]) -> Result<ResidentIndex, Error> {
        let hs: Vec<Handle> = leaves.iter().map(|m| m.to_handle()).collect();
        let hp: Vec<*mut RawKmerMinHash> = hs.iter().map(|h| h.0).collect();
        let raw = unsafe { smh_index_new(hp.as_ptr(), hp.len() as u32) };
        if raw.is_null() {
            return Err(take_error());
        }
        Ok(ResidentIndex { raw, _leaves: hs })
    }

    pub fn len(&self) -> usize {
        unsafe { smh_index_len(self.raw) as usize }
    }

    /// Positions of the leaves with similarity (or containment) above `threshold`, ascending.
    pub fn find(&self, query: &KmerMinHash, threshold: f64, containment: bool) -> Result<Vec<usize>, Error> {
        let q = query.to_handle();
        let mut out = vec![0u32; self.len().max(1)];
        let mut n = 0u32;
        check(unsafe { smh_index_find(self.raw, q.0, threshold, containment, out.as_mut_ptr(), &mut n) })?;
        Ok(out[..n as usize].iter().map(|&i| i as usize).collect())
    }
}

impl Drop for ResidentIndex {
    fn drop(&mut self) {
        unsafe { smh_index_free(self.raw) }
    }
}
