#!/bin/bash
# Applies the integration to a checkout of antoinedesbois/Ray-Tracer-Rust (the revision SURVEY.md describes: src/main.rs
# of 362 lines, render() at 242-317, thread fan-out at 275-303).  Written as line operations so that no line of the
# reference has to be quoted here.  Needs cargo + the crates of Cargo.lock, which this repository's build image lacks:
# it has never been run there (tests/test_ffi_layout.py is what keeps the binding honest without rustc).
#   integration/apply_to_reference.sh /path/to/Ray-Tracer-Rust /path/to/ray-tracer-rust_amd
set -euo pipefail
REF="${1:?path to the reference checkout}"
LIB="${2:?directory holding librtx.so}"
HERE="$(cd "$(dirname "$0")" && pwd)"
M="$REF/src/main.rs"
[ "$(wc -l < "$M")" = 362 ] || { echo "src/main.rs is not the 362-line revision this script was written against"; exit 1; }
cp "$HERE/rtx_ffi.rs" "$HERE/render_gpu.rs" "$REF/src/"
cp "$HERE/build.rs" "$REF/build.rs"
grep -q '^build' "$REF/Cargo.toml" || sed -i '/^\[package\]/a build = "build.rs"' "$REF/Cargo.toml"
# 1. the fan-out (spawn / render_pixel / put_pixel / recv), lines 275-303, becomes one call; the image is filled from
#    the returned rows.  `flat` is made in main() (step 3) and travels in a global the size of a pointer.
sed -i '275,303d' "$M"
sed -i '274r /dev/stdin' "$M" <<'RS'
    let table: Vec<rtx_ffi::Sample> = random_samples.iter().map(|s| rtx_ffi::Sample { s0: s.0, s1: s.1 }).collect();
    let flat = unsafe { &*FLAT.expect("main() flattens the primitives first") };
    let (rgb, stats) = render_gpu::render_frame(w, h, &scene.camera, &scene.light.primitives[0], flat, &table,
                                                NB_RAY, NB_LIGHT_SAMPLE).expect("librtx");
    {
        let mut im = img.lock().unwrap();
        for py in 0..h {
            for px in 0..w {
                let o = ((py * w + px) * 3) as usize;
                im.put_pixel(px, py, image::Rgba { data: [rgb[o], rgb[o + 1], rgb[o + 2], 255] });
            }
        }
    }
    println!("librtx: {} rays, {} primary hits, {:.3} ms on the device(s)", stats.rays, stats.primary_hits, stats.kernel_ms);
RS
# 2. module declarations and the global, after the crate's own `mod` lines at the top of the file
sed -i '0,/^mod /s//mod rtx_ffi;\nmod render_gpu;\nstatic mut FLAT: Option<*const render_gpu::FlatScene> = None;\nmod /' "$M"
# 3. main(): flatten the Vec<Primitive> just before BoundingVolumeHierarchy::new takes it by value
LINE=$(grep -n 'BoundingVolumeHierarchy::new(primitives)' "$M" | head -1 | cut -d: -f1)
sed -i "$((LINE - 1))r /dev/stdin" "$M" <<'RS'
    let flat = Box::new(render_gpu::flatten(&primitives));
    unsafe { FLAT = Some(Box::into_raw(flat) as *const render_gpu::FlatScene); }
RS
echo "patched $M; build with: RTX_LIB_DIR=$LIB cargo build --release"
