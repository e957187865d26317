//! src/rtx_ffi.rs — binding of librtx.so (include/rtx.h, RTX_ABI_VERSION 3) for the reference crate.
//!
//! Mirrors the header declaration by declaration.  The `#[repr(C)]` structs below are checked against the C header
//! WITHOUT a Rust toolchain: tests/test_ffi_layout.py parses this file, lays the fields out by the C rules, and
//! compares every offset and size with what a C compiler reports for include/rtx.h (integration/layout.json is the
//! table both sides must equal).  Edit the header, this file and the table together.
#![allow(dead_code)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const RTX_ABI_VERSION: c_int = 3;
pub const RTX_OK: c_int = 0;

/// One entry of the random-sample table.  `Vec<(f32, f32)>` (src/main.rs:253) has NO guaranteed layout — Rust tuples
/// are `repr(Rust)` — so the table handed to the library is a `Vec<Sample>` (or a `Vec<[f32; 2]>`): two packed f32.
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct Sample {
    pub s0: f32,
    pub s1: f32,
}

#[repr(C)]
pub struct RtxSceneDesc {
    pub width: u32,
    pub height: u32,
    pub eye: [f32; 3],
    pub u: [f32; 3],
    pub v: [f32; 3],
    pub w: [f32; 3],
    pub distance: f32,
    pub light_v0: [f32; 3],
    pub light_v1: [f32; 3],
    pub light_v2: [f32; 3],
    pub n_tris: u32,
    pub v0v1v2: *const f32,
    pub rgb: *const f32,
    pub tie_rank: *const u32,
    pub nb_ray: u32,
    pub nb_light_sample: u32,
    pub samples: *const f32,
    pub n_samples: u32,
    pub accel: u32,
    pub leaf_max: u32,
    pub reference_tree: u32,
    pub n_spheres: u32,
    pub spheres: *const f32,
    pub sphere_rgb: *const f32,
    pub kinds: *const u8,
}

#[repr(C)]
#[derive(Default)]
pub struct RtxStats {
    pub primary_rays: u64,
    pub primary_hits: u64,
    pub shadow_rays: u64,
    pub rays: u64,
    pub box_tests: u64,
    pub tri_tests: u64,
    pub wave_node_visits: u64,
    pub wave_tri_visits: u64,
    pub redo_tiles: u64,
    pub kernel_ms: f64,
    pub total_ms: f64,
}

#[repr(C)]
#[derive(Default)]
pub struct RtxSceneInfo {
    pub n_tris: u32,
    pub n_nodes: u32,
    pub n_leaves: u32,
    pub max_leaf_tris: u32,
    pub depth: u32,
    pub n_light_points: u32,
    pub n_ref_nodes: u32,
    pub n_global: u32,
    pub node_bytes: u64,
    pub tri_bytes: u64,
    pub shade_bytes: u64,
    pub sample_bytes: u64,
}

/// Opaque handle (library-owned).
#[repr(C)]
pub struct RtxScene {
    _private: [u8; 0],
}

extern "C" {
    pub fn rtx_abi_version() -> c_int;
    pub fn rtx_device_count() -> c_int;
    pub fn rtx_scene_create(desc: *const RtxSceneDesc, out: *mut *mut RtxScene) -> c_int;
    pub fn rtx_scene_destroy(scene: *mut RtxScene);
    pub fn rtx_scene_info(scene: *const RtxScene, info: *mut RtxSceneInfo) -> c_int;
    pub fn rtx_scene_upload(scene: *mut RtxScene, device: c_int) -> c_int;
    pub fn rtx_render_rows(scene: *mut RtxScene, device: c_int, row0: u32, nrows: u32, out_rgb: *mut u8,
                           stats: *mut RtxStats) -> c_int;
    pub fn rtx_render_frame(scene: *mut RtxScene, devices: *const c_int, n_devices: c_int, tile_rows: u32,
                            out_rgb: *mut u8, stats: *mut RtxStats) -> c_int;
    pub fn rtx_render_tiles_device(scene: *mut RtxScene, device: c_int, first_tile: u32, tile_stride: u32,
                                   tile_rows: u32, d_out_rgb: *mut c_void, d_out_bytes: usize, stream: *mut c_void,
                                   d_counters: *mut u64) -> c_int;
    pub fn rtx_tiles_rows(scene: *const RtxScene, first_tile: u32, tile_stride: u32, tile_rows: u32) -> u32;
    pub fn rtx_tiles_bytes(scene: *const RtxScene, first_tile: u32, tile_stride: u32, tile_rows: u32) -> usize;
    pub fn rtx_launch_timings(scene: *mut RtxScene, device: c_int, max_launches: c_int, schedule_ms: *mut f32,
                              shade_ms: *mut f32) -> c_int;
    pub fn rtx_strerror(err: c_int) -> *const c_char;
    pub fn rtx_last_hip_error() -> c_int;
    pub fn rtxh_scatter_tiles(frame: *mut u8, height: u32, width: u32, packed: *const u8, first_tile: u32,
                              tile_stride: u32, tile_rows: u32) -> c_int;
    pub fn rtxh_ref_leaf_rank(n_tris: u32, v0v1v2: *const f32, out_rank: *mut u32) -> c_int;
}

/// Error of a library call, with the library's own text.
pub fn check(rc: c_int) -> Result<(), String> {
    if rc == RTX_OK {
        Ok(())
    } else {
        let text = unsafe { CStr::from_ptr(rtx_strerror(rc)) };
        Err(format!("librtx: {} [{}]", text.to_string_lossy(), rc))
    }
}
