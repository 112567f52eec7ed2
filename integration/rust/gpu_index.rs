//! Safe wrappers over `ffi` (integration/rust/ffi.rs) for the reference's crate `vector_engine`:
//! `GpuFlatIndex` = one row-range shard on one GPU, `GpuShardedIndex` = one logical index over several GPUs of a node.
//! Written against the reference's own types (`DistanceMetric`, `VectorError`, `Result`, lib.rs:148-260); this image has
//! no Rust toolchain, so the file is not compiled here — the C ABI it calls is exercised by tests/ through ctypes and by
//! nmn_engine.cpp.
#![allow(unsafe_code)]
use std::ffi::{CStr, CString};
use std::path::Path;

use crate::ffi;
use crate::{DistanceMetric, Result, VectorError};

fn last_error() -> String {
    // thread-local in the library: the message of this thread's most recent failing call
    unsafe { CStr::from_ptr(ffi::nmn_last_error()) }.to_string_lossy().into_owned()
}

/// Status codes are the `VectorError` variants (include/neumann_gpu.h:32-50; lib.rs:148-200).
fn check(st: ffi::nmn_status, expected_dim: usize, got_dim: usize) -> Result<()> {
    match st {
        ffi::NMN_OK => Ok(()),
        ffi::NMN_ERR_DIMENSION_MISMATCH => Err(VectorError::DimensionMismatch { expected: expected_dim, got: got_dim }),
        ffi::NMN_ERR_EMPTY_VECTOR => Err(VectorError::EmptyVector),
        ffi::NMN_ERR_INVALID_TOP_K => Err(VectorError::InvalidTopK),
        ffi::NMN_ERR_CONFIGURATION => Err(VectorError::ConfigurationError(last_error())),
        ffi::NMN_ERR_IO => Err(VectorError::IoError(last_error())),
        ffi::NMN_ERR_SERIALIZATION => Err(VectorError::SerializationError(last_error())),
        _ => Err(VectorError::StorageError(last_error())),
    }
}

impl From<DistanceMetric> for ffi::nmn_metric {
    fn from(m: DistanceMetric) -> Self {
        match m {
            DistanceMetric::Cosine => ffi::nmn_metric::NMN_METRIC_COSINE,
            DistanceMetric::Euclidean => ffi::nmn_metric::NMN_METRIC_EUCLIDEAN,
            DistanceMetric::DotProduct => ffi::nmn_metric::NMN_METRIC_DOT_PRODUCT,
        }
    }
}

fn c_path(p: &Path) -> Result<CString> {
    CString::new(p.to_string_lossy().as_bytes()).map_err(|e| VectorError::IoError(e.to_string()))
}

pub struct GpuFlatIndex {
    raw: *mut ffi::nmn_index,
    dim: usize,
}
// One handle may be searched from any number of threads: callers that arrive while the shard is busy are queued and run
// together as ONE query batch (nmn_index_coalesce_stats); each gets exactly what it gets alone.  Writers drain the queue.
unsafe impl Send for GpuFlatIndex {}
unsafe impl Sync for GpuFlatIndex {}

impl GpuFlatIndex {
    /// `rows`: row-major n x dim, the vectors of the keys in `HnswCacheEntry` order (lib.rs:98).
    pub fn build(dim: usize, rows: &[f32], device: i32) -> Result<Self> {
        if dim == 0 {
            return Err(VectorError::EmptyVector);
        }
        let n = (rows.len() / dim) as u64;
        let desc = ffi::nmn_index_desc {
            dim: u32::try_from(dim).map_err(|_| VectorError::ConfigurationError("dimension exceeds u32".into()))?,
            flags: 0,
            capacity_rows: n + n / 4 + 1024, // spare rows for appends (Mirror in nmn_engine.cpp)
            row_base: 0,
            device,
            cand_cap: 0,
        };
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::nmn_index_create(&desc, &mut raw) }, dim, dim)?;
        let idx = Self { raw, dim };
        check(unsafe { ffi::nmn_index_upload(idx.raw, rows.as_ptr(), 0, n) }, dim, dim)?;
        Ok(idx)
    }

    pub fn len(&self) -> usize {
        unsafe { ffi::nmn_index_rows(self.raw) as usize }
    }

    /// Device memory the shard holds: (f32 rows, mirrors that exist right now, per-row factors), in bytes.  A shard keeps the
    /// f32 rows plus ONE approximate copy by default (int8 codes where the row stride is a multiple of 128 elements: 5 bytes per
    /// element in all); what `VectorEngine::gpu_memory_usage()` would sum over its `gpu_cache`.
    pub fn hbm_bytes(&self) -> (u64, u64, u64) {
        let (mut a, mut b, mut c) = (0u64, 0u64, 0u64);
        unsafe { ffi::nmn_index_hbm_bytes(self.raw, &mut a, &mut b, &mut c) };
        (a, b, c)
    }

    /// Same shape as `HNSWIndex::search` (tensor_store/src/hnsw.rs:2055): (row, score), best first; ties by row id.
    /// `mask`: bit i of word i/64 = row i takes part (pre-filter bitmap, the `live` bitmap of lazy deletes, or both ANDed).
    pub fn search(&self, q: &[f32], k: usize, metric: DistanceMetric, mask: Option<&[u64]>) -> Result<Vec<(usize, f32)>> {
        if q.len() != self.dim {
            return Err(VectorError::DimensionMismatch { expected: self.dim, got: q.len() });
        }
        let mut rows = vec![u64::MAX; k];
        let mut scores = vec![f32::NEG_INFINITY; k];
        let mut n = 0u32;
        let st = unsafe {
            ffi::nmn_index_search(
                self.raw, q.as_ptr(), 1, k as u32, metric.into(), mask.map_or(std::ptr::null(), |m| m.as_ptr()),
                rows.as_mut_ptr(), scores.as_mut_ptr(), &mut n, std::ptr::null_mut(),
            )
        };
        check(st, self.dim, q.len())?;
        Ok(rows.into_iter().zip(scores).take(n as usize).map(|(r, s)| (r as usize, s)).collect())
    }

    /// `nq` queries in one sweep of the shard (row-major nq x dim); result lists in query order.
    pub fn search_batch(&self, queries: &[f32], k: usize, metric: DistanceMetric) -> Result<Vec<Vec<(usize, f32)>>> {
        let nq = queries.len() / self.dim;
        let mut rows = vec![u64::MAX; nq * k];
        let mut scores = vec![f32::NEG_INFINITY; nq * k];
        let mut counts = vec![0u32; nq];
        let st = unsafe {
            ffi::nmn_index_search(
                self.raw, queries.as_ptr(), nq as u32, k as u32, metric.into(), std::ptr::null(), rows.as_mut_ptr(),
                scores.as_mut_ptr(), counts.as_mut_ptr(), std::ptr::null_mut(),
            )
        };
        check(st, self.dim, self.dim)?;
        Ok((0..nq)
            .map(|i| (0..counts[i] as usize).map(|j| (rows[i * k + j] as usize, scores[i * k + j])).collect())
            .collect())
    }

    /// Overwrite (or append at `len()`) one row: `store_embedding` keeps the mirror current instead of dropping it.
    pub fn set_row(&self, row: usize, v: &[f32]) -> Result<()> {
        if v.len() != self.dim {
            return Err(VectorError::DimensionMismatch { expected: self.dim, got: v.len() });
        }
        check(unsafe { ffi::nmn_index_set_row(self.raw, row as u64, v.as_ptr()) }, self.dim, v.len())
    }

    /// Device-layout snapshot (nmn_index_save): rows + magnitudes, checksummed; loads without re-deriving anything.
    pub fn save(&self, path: &Path) -> Result<()> {
        let p = c_path(path)?;
        check(unsafe { ffi::nmn_index_save(self.raw, p.as_ptr()) }, self.dim, self.dim)
    }

    /// `max_file_bytes` / `max_entries`: VectorEngineConfig::max_index_file_bytes / max_index_entries (lib.rs:644-646);
    /// 0 = no limit.  Exceeding either is a ConfigurationError with the reference's message.
    pub fn load(path: &Path, device: i32, max_file_bytes: u64, max_entries: u64) -> Result<Self> {
        let p = c_path(path)?;
        let ov = ffi::nmn_index_desc { dim: 0, flags: 0, capacity_rows: 0, row_base: 0, device, cand_cap: 0 };
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::nmn_index_load(p.as_ptr(), &ov, max_file_bytes, max_entries, &mut raw) }, 0, 0)?;
        let dim = unsafe { ffi::nmn_index_dim(raw) } as usize;
        Ok(Self { raw, dim })
    }
}

impl Drop for GpuFlatIndex {
    fn drop(&mut self) {
        unsafe { ffi::nmn_index_destroy(self.raw) };
    }
}

/// One logical index over the GPUs of a node (nmn_sharded_*): rows are split into equal contiguous ranges, every search
/// runs on all shards at once, the per-shard top-k blocks are gathered (RCCL all-gather over xGMI, or peer copies) and
/// merged on one device with ResultMerger's rule (query_router/src/distributed.rs:413-433: score desc, ties by row id).
pub struct GpuShardedIndex {
    raw: *mut ffi::nmn_sharded,
    dim: usize,
}
unsafe impl Send for GpuShardedIndex {}
unsafe impl Sync for GpuShardedIndex {}

impl GpuShardedIndex {
    pub fn build(dim: usize, rows: &[f32], devices: &[i32]) -> Result<Self> {
        let n = (rows.len() / dim.max(1)) as u64;
        let desc = ffi::nmn_sharded_desc {
            dim: dim as u32,
            flags: 0,
            capacity_rows: n,
            row_base: 0,
            n_shards: devices.len() as u32,
            gather: ffi::NMN_GATHER_AUTO,
            devices: devices.as_ptr(),
            cand_cap: 0,
            layout: ffi::NMN_SHARDED_LAYOUT_RANGES,
        };
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::nmn_sharded_create(&desc, &mut raw) }, dim, dim)?;
        let s = Self { raw, dim };
        check(unsafe { ffi::nmn_sharded_upload(s.raw, rows.as_ptr(), 0, n) }, dim, dim)?;
        Ok(s)
    }

    pub fn search(&self, q: &[f32], k: usize, metric: DistanceMetric, mask: Option<&[u64]>) -> Result<Vec<(usize, f32)>> {
        if q.len() != self.dim {
            return Err(VectorError::DimensionMismatch { expected: self.dim, got: q.len() });
        }
        let mut rows = vec![u64::MAX; k];
        let mut scores = vec![f32::NEG_INFINITY; k];
        let mut n = 0u32;
        let st = unsafe {
            ffi::nmn_sharded_search(
                self.raw, q.as_ptr(), 1, k as u32, metric.into(), mask.map_or(std::ptr::null(), |m| m.as_ptr()),
                rows.as_mut_ptr(), scores.as_mut_ptr(), &mut n, std::ptr::null_mut(),
            )
        };
        check(st, self.dim, q.len())?;
        Ok(rows.into_iter().zip(scores).take(n as usize).map(|(r, s)| (r as usize, s)).collect())
    }
}

impl Drop for GpuShardedIndex {
    fn drop(&mut self) {
        unsafe { ffi::nmn_sharded_destroy(self.raw) };
    }
}
