//! threecrate-hip: the normals + ICP path of threecrate on an AMD MI355X, behind the signatures of
//! threecrate-algorithms (`estimate_normals*`, `icp*`, `icp_point_to_point[_default]`, `icp_point_to_plane*`,
//! `multiscale_icp_point_to_point`, `gicp`, `kiss_icp`, `voxel_grid_filter`) and of the `threecrate-gpu` facade
//! (`gpu_estimate_normals`, `gpu_icp`, `gpu_icp_point_to_plane`, `gpu_batch_icp`, `gpu_voxel_grid_filter`,
//! `gpu_find_k_nearest[_batch]`, `gpu_find_radius_neighbors`; the reference's are `async fn`s around a wgpu queue, these
//! return when the result is there).  tests/test_abi_conformance.py checks that every name listed here has its `pub fn`
//! and that ffi.rs declares every `tc_*` export of the header.  Every function takes a [`HipContext`] (the role `GpuContext` plays in
//! threecrate-gpu: one device + one stream; not thread-safe, one context per thread / GPU).
//!
//! Layout facts the zero-copy calls rely on: `Point3f` = `nalgebra::Point3<f32>` is three contiguous f32
//! (threecrate-core/src/point.rs:8), `NormalPoint3f` is `#[repr(C)] { position, normal }` (point.rs:31-36),
//! `Isometry3<f32>` is passed as (qi, qj, qk, qw, tx, ty, tz).
pub mod ffi;

use nalgebra::{Isometry3, Quaternion, Translation3, UnitQuaternion};
use std::ffi::CStr;
use threecrate_algorithms::{GicpConfig, ICPResult, IcpScaleLevel, KissIcpConfig, MultiScaleIcpConfig, NormalEstimationConfig};
use threecrate_core::{Error, NearestNeighborSearch, NormalPoint3f, Point3f, PointCloud, Result, Vector3f};

/// One HIP device + stream + the library's grow-only device buffers (`tc_context`).
pub struct HipContext(*mut ffi::tc_context);

// the context is only ever used from the thread that owns the value
unsafe impl Send for HipContext {}

impl HipContext {
    /// `GpuContext::new` (threecrate-gpu/src/device.rs:16-50).  `Error::Gpu` when no MI355X is usable.
    pub fn new(device: i32) -> Result<Self> {
        let mut h: *mut ffi::tc_context = std::ptr::null_mut();
        let rc = unsafe { ffi::tc_context_create(device, &mut h) };
        if rc != ffi::TC_OK || h.is_null() {
            return Err(Error::Gpu(format!("no usable HIP device {device} (tc_status {rc})")));
        }
        Ok(HipContext(h))
    }

    fn check(&self, rc: i32) -> Result<()> {
        if rc == ffi::TC_OK {
            return Ok(());
        }
        let msg = unsafe {
            let p = ffi::tc_last_error_message(self.0);
            if p.is_null() { String::new() } else { CStr::from_ptr(p).to_string_lossy().into_owned() }
        };
        Err(match rc {                                   // threecrate-core/src/error.rs:7-28
            ffi::TC_INVALID_DATA => Error::InvalidData(msg),
            ffi::TC_ALGORITHM => Error::Algorithm(msg),
            ffi::TC_UNSUPPORTED => Error::Unsupported(msg),
            _ => Error::Gpu(msg),
        })
    }
}

impl Drop for HipContext {
    fn drop(&mut self) {
        unsafe { ffi::tc_context_destroy(self.0) }
    }
}

fn iso_to7(t: &Isometry3<f32>) -> [f32; 7] {
    let q = t.rotation.quaternion().coords;              // (i, j, k, w)
    [q[0], q[1], q[2], q[3], t.translation.x, t.translation.y, t.translation.z]
}

fn iso_from7(t: &[f32; 7]) -> Isometry3<f32> {
    Isometry3::from_parts(
        Translation3::new(t[4], t[5], t[6]),
        UnitQuaternion::new_unchecked(Quaternion::new(t[3], t[0], t[1], t[2])),   // Quaternion::new(w, i, j, k)
    )
}

fn xyz(cloud: &PointCloud<Point3f>) -> *const f32 {
    cloud.points.as_ptr() as *const f32
}

fn result_from(r: &ffi::tc_icp_result, corr: &[u32], n_valid: usize) -> ICPResult {
    ICPResult {
        transformation: iso_from7(&r.transformation),
        mse: r.mse,
        iterations: r.iterations as usize,
        converged: r.converged != 0,
        correspondences: corr[..n_valid].iter().enumerate().filter(|(_, &t)| t != u32::MAX).map(|(s, &t)| (s, t as usize)).collect(),
    }
}

/// `Option<f32>` -> the ABI's encoding (< 0 = None).  A negative `Some(d)` must not alias `None`: the reference rejects every
/// pair then (`distance > d` always holds, registration.rs:100-101) and fails with "Insufficient correspondences" -- AFTER its
/// own validation (registration.rs:266-276 / :517-531: empty clouds, the normals' length, max_iterations == 0 are InvalidData
/// first), which therefore runs here in the same order before the Algorithm error is returned.
fn encode_max_dist(d: Option<f32>, ns: usize, nt: usize, max_iters: usize, normals_len: Option<usize>) -> Result<f32> {
    match d {
        None => Ok(-1.0),
        Some(v) if v < 0.0 => {
            if ns == 0 || nt == 0 {
                return Err(Error::InvalidData("Source or target point cloud is empty".to_string()));
            }
            if let Some(nn) = normals_len {
                if nn != nt {
                    return Err(Error::InvalidData("target_normals length must equal the number of target points".to_string()));
                }
            }
            if max_iters == 0 {
                return Err(Error::InvalidData("Max iterations must be positive".to_string()));
            }
            Err(Error::Algorithm("Insufficient correspondences found".to_string()))
        }
        Some(v) => Ok(v),
    }
}

fn empty_result(corr: &mut Vec<u32>) -> ffi::tc_icp_result {
    ffi::tc_icp_result { transformation: [0.0; 7], mse: 0.0, iterations: 0, converged: 0, n_correspondences: 0, corr_target: corr.as_mut_ptr() }
}

// ---- normals (threecrate-algorithms/src/normals.rs:238-380) ------------------------------------------------

/// `estimate_normals_with_config` (normals.rs:257-260)
pub fn estimate_normals_with_config(ctx: &HipContext, cloud: &PointCloud<Point3f>, config: &NormalEstimationConfig)
    -> Result<PointCloud<NormalPoint3f>> {
    let n = cloud.points.len();
    let cfg = ffi::tc_normal_config {
        k_neighbors: config.k_neighbors as u64,
        radius: config.radius.unwrap_or(0.0),
        has_radius: config.radius.is_some() as i32,
        consistent_orientation: config.consistent_orientation as i32,
        has_viewpoint: config.viewpoint.is_some() as i32,
        viewpoint: config.viewpoint.map(|p| [p.x, p.y, p.z]).unwrap_or([0.0; 3]),
    };
    let mut out: Vec<NormalPoint3f> = Vec::with_capacity(n);
    ctx.check(unsafe { ffi::tc_estimate_normals(ctx.0, xyz(cloud), n, &cfg, out.as_mut_ptr() as *mut f32) })?;
    unsafe { out.set_len(n) };
    Ok(PointCloud::from_points(out))
}

/// `estimate_normals` (normals.rs:238-241)
pub fn estimate_normals(ctx: &HipContext, cloud: &PointCloud<Point3f>, k: usize) -> Result<PointCloud<NormalPoint3f>> {
    let config = NormalEstimationConfig { k_neighbors: k, ..Default::default() };
    estimate_normals_with_config(ctx, cloud, &config)
}

/// `estimate_normals_radius` (normals.rs:368-380)
pub fn estimate_normals_radius(ctx: &HipContext, cloud: &PointCloud<Point3f>, radius: f32, consistent_orientation: bool)
    -> Result<PointCloud<NormalPoint3f>> {
    let config = NormalEstimationConfig { k_neighbors: 10, radius: Some(radius), consistent_orientation, viewpoint: None };
    estimate_normals_with_config(ctx, cloud, &config)
}

// ---- ICP (threecrate-algorithms/src/registration.rs) ---------------------------------------------------------

/// `icp_detailed` (registration.rs:258-265)
pub fn icp_detailed(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, init: Isometry3<f32>,
                    max_iters: usize, max_correspondence_distance: Option<f32>, convergence_threshold: f32) -> Result<ICPResult> {
    let ns = source.points.len();
    let mut corr = vec![u32::MAX; ns.max(1)];
    let mut r = empty_result(&mut corr);
    let i7 = iso_to7(&init);
    ctx.check(unsafe {
        ffi::tc_icp_detailed(ctx.0, xyz(source), ns, xyz(target), target.points.len(), i7.as_ptr(), max_iters,
                             encode_max_dist(max_correspondence_distance, ns, target.points.len(), max_iters, None)?, convergence_threshold, &mut r)
    })?;
    Ok(result_from(&r, &corr, ns))
}

/// `icp` (registration.rs:232-242): any error returns `init`
pub fn icp(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, init: Isometry3<f32>, max_iters: usize)
    -> Isometry3<f32> {
    match icp_detailed(ctx, source, target, init, max_iters, None, 1e-6) {
        Ok(r) => r.transformation,
        Err(_) => init,
    }
}

/// `icp_point_to_point` (registration.rs:644-680)
pub fn icp_point_to_point(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, init: Isometry3<f32>,
                          max_iterations: usize, convergence_threshold: f32, max_correspondence_distance: Option<f32>) -> Result<ICPResult> {
    let ns = source.points.len();
    let mut corr = vec![u32::MAX; ns.max(1)];
    let mut r = empty_result(&mut corr);
    let i7 = iso_to7(&init);
    ctx.check(unsafe {
        ffi::tc_icp_point_to_point(ctx.0, xyz(source), ns, xyz(target), target.points.len(), i7.as_ptr(), max_iterations,
                                   convergence_threshold, encode_max_dist(max_correspondence_distance, ns, target.points.len(), max_iterations, None)?, &mut r)
    })?;
    Ok(result_from(&r, &corr, ns))
}

/// `icp_point_to_point_default` (registration.rs:694-701): threshold 1e-6, no correspondence cut-off
pub fn icp_point_to_point_default(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, init: Isometry3<f32>,
                                  max_iterations: usize) -> Result<ICPResult> {
    icp_point_to_point(ctx, source, target, init, max_iterations, 1e-6, None)
}

/// `multiscale_icp_point_to_point` (registration.rs:704-789): voxel-downsampled coarse-to-fine levels, then a final refinement on
/// the full clouds; the levels run on the device (tc_multiscale_icp_point_to_point), validation order and messages as the reference's.
/// A negative `Some(d)` cut-off of a level is passed through as `Some` (`f32::MIN_POSITIVE` keeps it from aliasing `None`; the level
/// then rejects every pair like the reference does).
pub fn multiscale_icp_point_to_point(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, init: Isometry3<f32>,
                                     config: &MultiScaleIcpConfig) -> Result<ICPResult> {
    let ns = source.points.len();
    let enc = |d: Option<f32>| match d { None => -1.0f32, Some(v) if v < 0.0 => f32::MIN_POSITIVE, Some(v) => v };
    let levels: Vec<ffi::tc_icp_scale_level> = config.levels.iter().map(|l: &IcpScaleLevel| ffi::tc_icp_scale_level {
        voxel_size: l.voxel_size, max_iterations: l.max_iterations, max_correspondence_distance: enc(l.max_correspondence_distance),
    }).collect();
    let cfg = ffi::tc_multiscale_icp_config {
        levels: levels.as_ptr(), n_levels: levels.len(),
        final_refinement_iterations: config.final_refinement_iterations,
        final_max_correspondence_distance: enc(config.final_max_correspondence_distance),
        convergence_threshold: config.convergence_threshold,
    };
    let mut corr = vec![u32::MAX; ns.max(1)];
    let mut r = empty_result(&mut corr);
    let i7 = iso_to7(&init);
    ctx.check(unsafe { ffi::tc_multiscale_icp_point_to_point(ctx.0, xyz(source), ns, xyz(target), target.points.len(), i7.as_ptr(), &cfg, &mut r) })?;
    Ok(result_from(&r, &corr, ns))
}

/// `icp_point_to_plane_detailed` (registration.rs:508-516)
pub fn icp_point_to_plane_detailed(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>,
                                   target_normals: &[Vector3f], init: Isometry3<f32>, max_iters: usize,
                                   max_correspondence_distance: Option<f32>, convergence_threshold: f32) -> Result<ICPResult> {
    let ns = source.points.len();
    let mut corr = vec![u32::MAX; ns.max(1)];
    let mut r = empty_result(&mut corr);
    let i7 = iso_to7(&init);
    ctx.check(unsafe {
        ffi::tc_icp_point_to_plane_detailed(ctx.0, xyz(source), ns, xyz(target), target.points.len(),
                                            target_normals.as_ptr() as *const f32, target_normals.len(), 3, i7.as_ptr(), max_iters,
                                            encode_max_dist(max_correspondence_distance, ns, target.points.len(), max_iters, Some(target_normals.len()))?,
                                            convergence_threshold, &mut r)
    })?;
    Ok(result_from(&r, &corr, ns))
}

/// `icp_point_to_plane` (registration.rs:488-494)
pub fn icp_point_to_plane(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, target_normals: &[Vector3f],
                          init: Isometry3<f32>, max_iters: usize) -> Result<ICPResult> {
    icp_point_to_plane_detailed(ctx, source, target, target_normals, init, max_iters, None, 1e-6)
}

/// `gicp` (gicp.rs:100-105)
pub fn gicp(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, init: Isometry3<f32>, config: GicpConfig)
    -> Result<ICPResult> {
    let ns = source.points.len();
    let mut corr = vec![u32::MAX; ns.max(1)];
    let mut r = empty_result(&mut corr);
    let i7 = iso_to7(&init);
    let cfg = ffi::tc_gicp_config {
        max_iterations: config.max_iterations,
        max_correspondence_distance: config.max_correspondence_distance,
        convergence_threshold: config.convergence_threshold,
        k_correspondences: config.k_correspondences,
    };
    ctx.check(unsafe { ffi::tc_gicp(ctx.0, xyz(source), ns, xyz(target), target.points.len(), i7.as_ptr(), &cfg, &mut r) })?;
    Ok(result_from(&r, &corr, ns))
}

/// `kiss_icp` (kiss_icp.rs:183-188).  Correspondence source indices refer to the voxel-downsampled source.
pub fn kiss_icp(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, init: Isometry3<f32>, config: KissIcpConfig)
    -> Result<ICPResult> {
    let ns = source.points.len();
    let mut corr = vec![u32::MAX; ns.max(1)];
    let mut r = empty_result(&mut corr);
    let i7 = iso_to7(&init);
    let cfg = ffi::tc_kiss_icp_config { voxel_size: config.voxel_size, max_range: config.max_range, min_range: config.min_range,
                                        max_iterations: config.max_iterations };
    let mut n_down = 0usize;
    ctx.check(unsafe { ffi::tc_kiss_icp(ctx.0, xyz(source), ns, xyz(target), target.points.len(), i7.as_ptr(), &cfg, &mut r, &mut n_down) })?;
    Ok(result_from(&r, &corr, n_down))
}

// ---- filters / neighbour search ----------------------------------------------------------------------------

/// `voxel_grid_filter` (filtering.rs:38-133); voxels come out sorted by (kx, ky, kz)
pub fn voxel_grid_filter(ctx: &HipContext, cloud: &PointCloud<Point3f>, voxel_size: f32) -> Result<PointCloud<Point3f>> {
    let n = cloud.points.len();
    let mut out: Vec<Point3f> = Vec::with_capacity(n.max(1));
    let mut n_out = 0usize;
    ctx.check(unsafe { ffi::tc_voxel_grid_filter(ctx.0, xyz(cloud), n, voxel_size, out.as_mut_ptr() as *mut f32, &mut n_out) })?;
    unsafe { out.set_len(n_out) };
    Ok(PointCloud::from_points(out))
}

/// `gpu_find_k_nearest_batch` (threecrate-gpu/src/nearest_neighbor.rs:345-355)
pub fn find_k_nearest_batch(ctx: &HipContext, points: &[Point3f], queries: &[Point3f], k: usize) -> Result<Vec<Vec<(usize, f32)>>> {
    let nq = queries.len();
    let kk = k.max(1);
    let (mut idx, mut dist, mut cnt) = (vec![0u32; nq * kk], vec![0f32; nq * kk], vec![0u32; nq.max(1)]);
    ctx.check(unsafe {
        ffi::tc_knn(ctx.0, points.as_ptr() as *const f32, points.len(), queries.as_ptr() as *const f32, nq, k, idx.as_mut_ptr(),
                    dist.as_mut_ptr(), cnt.as_mut_ptr())
    })?;
    Ok((0..nq).map(|q| (0..cnt[q] as usize).map(|j| (idx[q * kk + j] as usize, dist[q * kk + j])).collect()).collect())
}

/// longest neighbour list of the device k-NN kernels (threecrate_hip.h, "Limits of this backend")
pub const KNN_MAX_K: usize = 2047;

/// The `KdTree` of this backend: built once, queried many times (`KdTree::new`, nearest_neighbor.rs:37-58).
/// Implements `threecrate_core::NearestNeighborSearch` (core/traits.rs:6-12), so it drops into code that is generic
/// over the trait.  Borrows the context: destroy order is enforced by the lifetime.
pub struct HipKdTree<'a> {
    ctx: &'a HipContext,
    raw: *mut ffi::tc_search_index,
}

impl<'a> HipKdTree<'a> {
    pub fn new(ctx: &'a HipContext, points: &[Point3f]) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        ctx.check(unsafe { ffi::tc_search_index_create(ctx.0, points.as_ptr() as *const f32, points.len(), 16, &mut raw) })?;
        Ok(Self { ctx, raw })
    }

    pub fn len(&self) -> usize { unsafe { ffi::tc_search_index_size(self.raw) } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }

    fn query(&self, queries: &[Point3f], k: usize, radius: f32) -> Result<Vec<Vec<(usize, f32)>>> {
        let nq = queries.len();
        let kk = k.max(1);
        let (mut idx, mut dist, mut cnt) = (vec![0u32; nq * kk], vec![0f32; nq * kk], vec![0u32; nq.max(1)]);
        self.ctx.check(unsafe {
            ffi::tc_search_index_query(self.raw, queries.as_ptr() as *const f32, nq, k, radius, idx.as_mut_ptr(), dist.as_mut_ptr(), cnt.as_mut_ptr())
        })?;
        Ok((0..nq).map(|q| (0..cnt[q] as usize).map(|j| (idx[q * kk + j] as usize, dist[q * kk + j])).collect()).collect())
    }

    /// many queries in one launch
    pub fn find_k_nearest_batch(&self, queries: &[Point3f], k: usize) -> Result<Vec<Vec<(usize, f32)>>> {
        self.query(queries, k, -1.0)
    }
}

impl NearestNeighborSearch for HipKdTree<'_> {
    /// `KdTree::find_k_nearest` (nearest_neighbor.rs:177-251): the `min(k, len())` nearest, ascending.  The device list holds up to
    /// `KNN_MAX_K` = 2047 entries (threecrate_hip.h "Limits of this backend"; a larger k is TC_UNSUPPORTED at the ABI).  The trait
    /// returns a plain Vec and the reference's tree has no cap, so a larger k is served from radius sets: the k nearest are the
    /// first k of any ball that holds at least k records -- the radius starts at the 2047th neighbour's distance scaled by
    /// cbrt(k / 2047) and doubles until the count reaches k (at most a few rounds; `f32::MAX` covers every finite cloud).
    fn find_k_nearest(&self, query: &Point3f, k: usize) -> Vec<(usize, f32)> {
        let k = k.min(self.len());
        if k == 0 { return Vec::new(); }
        if k <= KNN_MAX_K {
            return self.query(std::slice::from_ref(query), k, -1.0).map(|mut v| v.remove(0)).unwrap_or_default();
        }
        let seed = match self.query(std::slice::from_ref(query), KNN_MAX_K, -1.0) { Ok(mut v) => v.remove(0), Err(_) => return Vec::new() };
        let r0 = seed.last().map(|e| e.1).unwrap_or(0.0);
        let mut r = if r0 > 0.0 && r0.is_finite() { r0 * (k as f32 / KNN_MAX_K as f32).cbrt() * 1.05 } else { 1.0e-6 };
        let q = query as *const Point3f as *const f32;
        loop {
            let mut count = 0u32;
            if self.ctx.check(unsafe { ffi::tc_search_index_radius_count(self.raw, q, 1, r, &mut count) }).is_err() { return Vec::new(); }
            if count as usize >= k || r >= f32::MAX { break; }
            r = if r < f32::MAX / 2.0 { r * 2.0 } else { f32::MAX };
        }
        let mut out = self.find_radius_neighbors(query, r);          // ascending by distance
        out.truncate(k);
        out
    }

    /// every neighbour within `radius`, nearest first (nearest_neighbor.rs:254-298): count, then fill
    fn find_radius_neighbors(&self, query: &Point3f, radius: f32) -> Vec<(usize, f32)> {
        if !(radius > 0.0) || self.is_empty() { return Vec::new(); }
        let q = query as *const Point3f as *const f32;
        let mut count = 0u32;
        if self.ctx.check(unsafe { ffi::tc_search_index_radius_count(self.raw, q, 1, radius, &mut count) }).is_err() { return Vec::new(); }
        let total = count as usize;
        let (mut idx, mut dist, offsets) = (vec![0u32; total], vec![0f32; total], [0u64]);
        if total > 0 && self.ctx.check(unsafe {
            ffi::tc_search_index_radius_fill(self.raw, q, 1, radius, offsets.as_ptr(), total, idx.as_mut_ptr(), dist.as_mut_ptr())
        }).is_err() { return Vec::new(); }
        let mut out: Vec<(usize, f32)> = idx.into_iter().map(|i| i as usize).zip(dist).collect();
        out.sort_by(|a, b| a.1.partial_cmp(&b.1).unwrap_or(std::cmp::Ordering::Equal));
        out
    }
}

impl Drop for HipKdTree<'_> {
    fn drop(&mut self) { unsafe { ffi::tc_search_index_destroy(self.raw) } }
}

// ---- threecrate-gpu facade (gpu/normals.rs:443-447, gpu/icp.rs:977-1025) ---------------------------------------

/// `gpu_icp(&ctx, source, target, max_iterations, convergence_threshold, max_correspondence_distance)`
pub fn gpu_icp(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, max_iterations: usize,
               convergence_threshold: f32, max_correspondence_distance: f32) -> Result<Isometry3<f32>> {
    Ok(icp_point_to_point(ctx, source, target, Isometry3::identity(), max_iterations, convergence_threshold,
                          Some(max_correspondence_distance))?.transformation)
}

/// `gpu_estimate_normals(&ctx, &mut cloud, k)`
pub fn gpu_estimate_normals(ctx: &HipContext, cloud: &mut PointCloud<Point3f>, k: usize) -> Result<PointCloud<NormalPoint3f>> {
    estimate_normals(ctx, cloud, k)
}

/// `GpuPointToPlaneICPResult` (threecrate-gpu/src/icp.rs:821-828)
#[derive(Debug, Clone)]
pub struct GpuPointToPlaneICPResult {
    pub transformation: Isometry3<f32>,
    pub final_error: f32,
    pub iterations: usize,
    pub converged: bool,
}

/// `gpu_icp_point_to_plane(&ctx, source, target, target_normals, max_iterations, convergence_threshold,
/// max_correspondence_distance)` (threecrate-gpu/src/icp.rs:1017-1036); starts from the identity like the reference's.
pub fn gpu_icp_point_to_plane(ctx: &HipContext, source: &PointCloud<Point3f>, target: &PointCloud<Point3f>, target_normals: &[Vector3f],
                              max_iterations: usize, convergence_threshold: f32, max_correspondence_distance: f32)
    -> Result<GpuPointToPlaneICPResult> {
    let r = icp_point_to_plane_detailed(ctx, source, target, target_normals, Isometry3::identity(), max_iterations,
                                        Some(max_correspondence_distance), convergence_threshold)?;
    Ok(GpuPointToPlaneICPResult { transformation: r.transformation, final_error: r.mse, iterations: r.iterations, converged: r.converged })
}

/// `BatchICPJob` (threecrate-gpu/src/icp.rs:132-139)
#[derive(Debug, Clone)]
pub struct BatchICPJob {
    pub source: PointCloud<Point3f>,
    pub target: PointCloud<Point3f>,
    pub max_iterations: usize,
    pub convergence_threshold: f32,
    pub max_correspondence_distance: f32,
}

/// `BatchICPResult` (threecrate-gpu/src/icp.rs:142-147)
#[derive(Debug, Clone)]
pub struct BatchICPResult {
    pub transformation: Isometry3<f32>,
    pub final_error: f32,
    pub iterations: usize,
}

/// `gpu_batch_icp(&ctx, &jobs)` (threecrate-gpu/src/icp.rs:997-1002, semantics :151-185): independent pairs, each from the
/// identity.  Job i runs on `contexts[i % contexts.len()]` -- one context per GPU of the node gives one pair per GPU
/// (tc_batch_icp: a host thread per context).  A job that fails returns its error for the whole batch, like the reference's `?`.
pub fn gpu_batch_icp(contexts: &[&HipContext], jobs: &[BatchICPJob]) -> Result<Vec<BatchICPResult>> {
    if contexts.is_empty() {
        return Err(Error::InvalidData("gpu_batch_icp needs at least one context".to_string()));
    }
    let raw_ctx: Vec<*mut ffi::tc_context> = contexts.iter().map(|c| c.0).collect();
    let cj: Vec<ffi::tc_batch_icp_job> = jobs.iter().map(|j| ffi::tc_batch_icp_job {
        source: xyz(&j.source), n_source: j.source.points.len(), target: xyz(&j.target), n_target: j.target.points.len(),
        max_iterations: j.max_iterations, convergence_threshold: j.convergence_threshold,
        max_correspondence_distance: j.max_correspondence_distance,
    }).collect();
    let mut cr = vec![ffi::tc_batch_icp_result { transformation: [0.0; 7], final_error: 0.0, iterations: 0, status: ffi::TC_OK }; jobs.len()];
    let rc = unsafe { ffi::tc_batch_icp(raw_ctx.as_ptr(), raw_ctx.len(), cj.as_ptr(), cj.len(), cr.as_mut_ptr()) };
    if rc != ffi::TC_OK {
        return Err(Error::InvalidData("tc_batch_icp: bad arguments".to_string()));
    }
    cr.iter().enumerate().map(|(i, r)| {
        contexts[i % contexts.len()].check(r.status)?;
        Ok(BatchICPResult { transformation: iso_from7(&r.transformation), final_error: r.final_error, iterations: r.iterations as usize })
    }).collect()
}

/// `gpu_voxel_grid_filter(&ctx, &cloud, voxel_size)` (threecrate-gpu/src/filtering.rs:908-917) with the CPU filter's semantics
/// (centroid per occupied voxel, filtering.rs:38-133 -- not the wgpu shader's hash-bucket "first point wins")
pub fn gpu_voxel_grid_filter(ctx: &HipContext, cloud: &PointCloud<Point3f>, voxel_size: f32) -> Result<PointCloud<Point3f>> {
    voxel_grid_filter(ctx, cloud, voxel_size)
}

/// `gpu_find_k_nearest(&ctx, points, &query, k)` (threecrate-gpu/src/nearest_neighbor.rs:332-342)
pub fn gpu_find_k_nearest(ctx: &HipContext, points: &[Point3f], query: &Point3f, k: usize) -> Result<Vec<(usize, f32)>> {
    Ok(find_k_nearest_batch(ctx, points, std::slice::from_ref(query), k)?.into_iter().next().unwrap_or_default())
}

/// `gpu_find_k_nearest_batch(&ctx, points, query_points, k)` (threecrate-gpu/src/nearest_neighbor.rs:345-355)
pub fn gpu_find_k_nearest_batch(ctx: &HipContext, points: &[Point3f], query_points: &[Point3f], k: usize) -> Result<Vec<Vec<(usize, f32)>>> {
    find_k_nearest_batch(ctx, points, query_points, k)
}

/// `gpu_find_radius_neighbors(&ctx, points, &query, radius)` (threecrate-gpu/src/nearest_neighbor.rs:358-367): the reference asks
/// its kernel for the 32 nearest and keeps those within `radius`; same rule here (tc_radius_search with k_max = 32), nearest first.
pub fn gpu_find_radius_neighbors(ctx: &HipContext, points: &[Point3f], query: &Point3f, radius: f32) -> Result<Vec<(usize, f32)>> {
    const K_MAX: usize = 32;
    let (mut idx, mut dist, mut cnt) = (vec![0u32; K_MAX], vec![0f32; K_MAX], [0u32; 1]);
    ctx.check(unsafe {
        ffi::tc_radius_search(ctx.0, points.as_ptr() as *const f32, points.len(), query as *const Point3f as *const f32, 1, radius, K_MAX,
                              idx.as_mut_ptr(), dist.as_mut_ptr(), cnt.as_mut_ptr())
    })?;
    Ok((0..cnt[0] as usize).map(|j| (idx[j] as usize, dist[j])).collect())
}


// ---- device-resident cloud handles (include/threecrate_hip.h "tc_cloud": SURVEY.md 8b) ---------------------------------

/// A cloud that lives in HBM, is indexed once and keeps its normals in the layout the ICP kernels read.
/// `let prev = HipCloud::upload(&ctx, &frame0)?; prev.estimate_normals(16)?; let cur = HipCloud::upload(&ctx, &frame1)?;
///  let r = cur.icp_point_to_plane(&prev, Isometry3::identity(), 50, None, 1e-6)?;`
pub struct HipCloud<'a> { raw: *mut ffi::tc_cloud, ctx: &'a HipContext }

impl<'a> HipCloud<'a> {
    pub fn upload(ctx: &'a HipContext, cloud: &PointCloud<Point3f>) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        ctx.check(unsafe { ffi::tc_cloud_upload(ctx.0, xyz(cloud), cloud.points.len(), &mut raw) })?;
        Ok(HipCloud { raw, ctx })
    }
    pub fn len(&self) -> usize { unsafe { ffi::tc_cloud_size(self.raw) } }
    pub fn is_empty(&self) -> bool { self.len() == 0 }
    /// `estimate_normals` (normals.rs:238-241); the normals stay with the handle, nothing is copied back
    pub fn estimate_normals(&self, k: usize) -> Result<()> {
        let mut cfg = unsafe { std::mem::zeroed::<ffi::tc_normal_config>() };
        unsafe { ffi::tc_normal_config_default(&mut cfg) };
        cfg.k_neighbors = k as u64;
        self.ctx.check(unsafe { ffi::tc_cloud_estimate_normals(self.raw, &cfg, std::ptr::null_mut()) })
    }
    /// `icp_point_to_plane_detailed` (registration.rs:508-516) with `self` as the source and the target handle's normals
    pub fn icp_point_to_plane(&self, target: &HipCloud, init: Isometry3<f32>, max_iters: usize, max_correspondence_distance: Option<f32>,
                              convergence_threshold: f32) -> Result<ICPResult> {
        let mut none = Vec::new();
        let mut r = empty_result(&mut none);
        r.corr_target = std::ptr::null_mut();            // (device memory when wanted: see the header)
        let i7 = iso_to7(&init);
        self.ctx.check(unsafe {
            ffi::tc_cloud_icp_point_to_plane(self.raw, target.raw, i7.as_ptr(), max_iters,
                                             encode_max_dist(max_correspondence_distance, self.len(), target.len(), max_iters, None)?,
                                             convergence_threshold, &mut r)
        })?;
        Ok(result_from(&r, &[], 0))
    }
}

impl Drop for HipCloud<'_> {
    fn drop(&mut self) { unsafe { ffi::tc_cloud_destroy(self.raw) } }
}

// ---- one registration over the GPUs of a node: one process or thread per GPU ------------------------------

/// RCCL communicator of this rank, bound to the context's stream.  Rank 0 calls `HipComm::unique_id()`, the host distributes the
/// 128 bytes (pipe, file, MPI ...), every rank calls `HipComm::create`.
pub struct HipComm<'a> { raw: *mut ffi::tc_comm, ctx: &'a HipContext }

impl<'a> HipComm<'a> {
    pub fn unique_id() -> Result<[u8; ffi::TC_COMM_ID_BYTES]> {
        let mut id = [0u8; ffi::TC_COMM_ID_BYTES];
        match unsafe { ffi::tc_comm_unique_id(id.as_mut_ptr()) } {
            ffi::TC_OK => Ok(id),
            _ => Err(Error::Unsupported("RCCL is not available to libthreecrate_hip".to_string())),
        }
    }
    pub fn create(ctx: &'a HipContext, nranks: i32, rank: i32, id: &[u8; ffi::TC_COMM_ID_BYTES]) -> Result<Self> {
        let mut raw = std::ptr::null_mut();
        ctx.check(unsafe { ffi::tc_comm_create(ctx.0, nranks, rank, id.as_ptr(), &mut raw) })?;
        Ok(HipComm { raw, ctx })
    }
    pub fn rank(&self) -> i32 { unsafe { ffi::tc_comm_rank(self.raw) } }
    pub fn size(&self) -> i32 { unsafe { ffi::tc_comm_size(self.raw) } }

    /// `icp_point_to_plane_detailed` of ONE cloud pair over all ranks: every rank passes the same device-resident clouds
    /// (pointers to n x 3 floats in HBM), the library takes this rank's index range of the source (TC_SHARD_INDEX: each rank
    /// orders n / W points by target cell) and runs one ncclAllReduce of the packed 6x6 system per iteration on the compute
    /// stream.  Every rank returns the same result.
    #[allow(clippy::too_many_arguments)]
    pub unsafe fn icp_point_to_plane_device(&self, d_source: *const f32, n_source: usize, d_target: *const f32, n_target: usize,
                                            d_target_normals: *const f32, normal_stride: usize, init: Isometry3<f32>, max_iters: usize,
                                            max_correspondence_distance: Option<f32>, convergence_threshold: f32) -> Result<ICPResult> {
        let mut none = Vec::new();
        let mut r = empty_result(&mut none);
        r.corr_target = std::ptr::null_mut();
        let i7 = iso_to7(&init);
        self.ctx.check(ffi::tc_sharded_icp_point_to_plane_device(self.ctx.0, self.raw, ffi::TC_SHARD_INDEX, d_source, n_source, d_target,
                                                                  n_target, d_target_normals, n_target, normal_stride, i7.as_ptr(), max_iters,
                                                                  encode_max_dist(max_correspondence_distance, n_source, n_target, max_iters, None)?,
                                                                  convergence_threshold, &mut r))?;
        Ok(result_from(&r, &[], 0))
    }
}

impl Drop for HipComm<'_> {
    fn drop(&mut self) { unsafe { ffi::tc_comm_destroy(self.raw) } }
}
