//! MI355X backend for leann-rs — UNCOMPILED source a maintainer would add as `src/backend/gpu.rs` (no Rust toolchain
//! exists in the build image of leann-rs_amd).  Implements the same trait as HnswSearcher /
//! DiskAnnSearcher (src/backend/traits.rs:11-30) over `libleann_hip.so` (include/leann_backend.h).
//!
//! Wiring:
//!   src/backend/mod.rs:23-45   load_searcher:  Hnsw => Box::new(gpu::GpuSearcher::load(index_path, dimensions, 0)?),
//!                                              DiskAnn => Box::new(gpu::GpuSearcher::load(index_path, dimensions, 1)?)
//!   src/backend/mod.rs:55-100  BackendBuilder::{build, add_to_index} => gpu::build / gpu::add_to_index
//!   build.rs                   cargo:rustc-link-search=native=<…>/leann-rs_amd/csrc ; cargo:rustc-link-lib=dylib=leann_hip
//!
//! LEANN_DEVICES: "0" (default) or a list / range ("0-7"): the index is then sharded over those GPUs behind the same handle.
//! LEANN_LOG=info: which file path was taken (own format, cached GPU graph, rebuild from documents.embeddings).
use std::ffi::{c_char, c_int, CStr, CString};
use std::path::Path;

use super::traits::BackendSearcher;

#[repr(C)]
pub struct LeannBackend {
    _p: [u8; 0],
}

#[link(name = "leann_hip")]
extern "C" {
    fn leann_last_error() -> *const c_char;
    fn leann_backend_open(stem: *const c_char, backend: c_int, dims: usize, device_spec: *const c_char,
                          out: *mut *mut LeannBackend) -> c_int;
    fn leann_backend_search(h: *const LeannBackend, query: *const f32, top_k: usize, complexity: usize,
                            keys: *mut u64, dists: *mut f32, n_out: *mut usize) -> c_int;
    #[allow(dead_code)]
    fn leann_backend_set_coalescing(h: *mut LeannBackend, wait_us: u32, max_batch: u32) -> c_int;
    fn leann_backend_len(h: *const LeannBackend) -> usize;
    fn leann_backend_close(h: *mut LeannBackend);
    fn leann_backend_build(backend: c_int, vectors: *const f32, n: usize, dims: usize, graph_degree: usize,
                           complexity: usize, stem: *const c_char) -> c_int;
    fn leann_backend_add(backend: c_int, vectors: *const f32, n: usize, dims: usize, start_id: usize,
                         stem: *const c_char) -> c_int;
}

fn last_error() -> anyhow::Error {
    let s = unsafe { CStr::from_ptr(leann_last_error()) }.to_string_lossy().into_owned();
    anyhow::anyhow!(s)
}

pub struct GpuSearcher {
    h: *mut LeannBackend,
}
// leann_backend_search is re-entrant on one handle (per-call workspace + stream inside the library)
unsafe impl Send for GpuSearcher {}
unsafe impl Sync for GpuSearcher {}

impl GpuSearcher {
    /// replaces HnswSearcher::load (hnsw.rs:18-75) / DiskAnnSearcher::load (diskann.rs:21-43); `backend`: 0 = hnsw, 1 = diskann
    pub fn load(index_path: &Path, dimensions: usize, backend: c_int) -> anyhow::Result<Self> {
        let stem = CString::new(index_path.to_string_lossy().as_bytes())?;
        let dev = CString::new(std::env::var("LEANN_DEVICES").unwrap_or_default())?;
        let mut h = std::ptr::null_mut();
        let rc = unsafe { leann_backend_open(stem.as_ptr(), backend, dimensions, dev.as_ptr(), &mut h) };
        if rc != 0 {
            return Err(last_error());
        }
        // a server calls search() from many tokio workers (cli/serve.rs:289-292): the library batches concurrent callers by itself
        // (a lone caller — `leann search` — is answered directly); leann_backend_set_coalescing(h, wait_us, max_batch) only tunes that
        Ok(Self { h })
    }
}

impl BackendSearcher for GpuSearcher {
    fn search(&self, query: &[f32], top_k: usize, complexity: usize) -> anyhow::Result<(Vec<u64>, Vec<f32>)> {
        let (mut keys, mut dists, mut n) = (vec![0u64; top_k], vec![0f32; top_k], 0usize);
        let rc = unsafe {
            leann_backend_search(self.h, query.as_ptr(), top_k, complexity, keys.as_mut_ptr(), dists.as_mut_ptr(), &mut n)
        };
        if rc != 0 {
            return Err(last_error());
        }
        keys.truncate(n);
        dists.truncate(n);
        Ok((keys, dists))
    }

    fn len(&self) -> usize {
        unsafe { leann_backend_len(self.h) }
    }
}

impl Drop for GpuSearcher {
    fn drop(&mut self) {
        unsafe { leann_backend_close(self.h) }
    }
}

fn flatten(embeddings: &[Vec<f32>], dimensions: usize) -> anyhow::Result<Vec<f32>> {
    let mut flat = Vec::with_capacity(embeddings.len() * dimensions);
    for e in embeddings {
        anyhow::ensure!(e.len() == dimensions, "Embedding dimension mismatch: expected {}, got {}", dimensions, e.len());
        flat.extend_from_slice(e);
    }
    Ok(flat)
}

/// BackendBuilder::build (mod.rs:55-79): the graph is built on the GPU and written as "<stem>.index" / ".diskann"
pub fn build(backend: c_int, embeddings: &[Vec<f32>], index_path: &Path, dimensions: usize, graph_degree: usize,
             complexity: usize) -> anyhow::Result<()> {
    let flat = flatten(embeddings, dimensions)?;
    let stem = CString::new(index_path.to_string_lossy().as_bytes())?;
    let rc = unsafe {
        leann_backend_build(backend, flat.as_ptr(), embeddings.len(), dimensions, graph_degree, complexity, stem.as_ptr())
    };
    if rc != 0 {
        return Err(last_error());
    }
    Ok(())
}

/// BackendBuilder::add_to_index (mod.rs:82-100); DiskANN answers with the reference's "does not support incremental updates"
pub fn add_to_index(backend: c_int, embeddings: &[Vec<f32>], index_path: &Path, dimensions: usize, start_id: usize)
                    -> anyhow::Result<()> {
    let flat = flatten(embeddings, dimensions)?;
    let stem = CString::new(index_path.to_string_lossy().as_bytes())?;
    let rc = unsafe { leann_backend_add(backend, flat.as_ptr(), embeddings.len(), dimensions, start_id, stem.as_ptr()) };
    if rc != 0 {
        return Err(last_error());
    }
    Ok(())
}
