#!/bin/bash
# Applies the integration to a checkout of antoinedesbois/Ray-Tracer-Rust (the revision SURVEY.md describes: src/main.rs
# of 364 lines by `wc -l`, render() at 242-317, thread fan-out at 275-303, `let scene = Scene {` at 349).  Written as
# line operations located by short anchors, so that no line of the reference has to be quoted here.
#
# Building the result needs cargo + the crates of Cargo.lock, which this repository's build image lacks; what IS run
# there, on a temporary copy of the reference, is this script (tests/test_integration_patch.py: exit status, brace
# balance, order of the inserted statements against the moves of `scene` and `random_samples`, every rtx_ffi:: /
# render_gpu:: item the inserted code names exists).
#   integration/apply_to_reference.sh /path/to/Ray-Tracer-Rust /path/to/ray-tracer-rust_amd
set -euo pipefail
REF="${1:?path to the reference checkout}"
LIB="${2:?directory holding librtx.so}"
HERE="$(cd "$(dirname "$0")" && pwd)"
M="$REF/src/main.rs"
die() { echo "apply_to_reference.sh: $*" >&2; exit 1; }
line_of() {   # line number of the only line matching $1, or die
    local n; n=$(grep -n -- "$1" "$M" | cut -d: -f1 || true)
    [ "$(echo "$n" | wc -w)" = 1 ] || die "anchor '$1' matches $(echo "$n" | wc -w) lines of src/main.rs, expected 1"
    echo "$n"
}
[ -f "$M" ] || die "$M not found"
grep -q 'mod rtx_ffi;' "$M" && die "src/main.rs is already patched"

# anchors of the revision this was written against; every one must be where SURVEY.md says it is
CHAN=$(line_of 'mpsc::channel()')                      # 250: the completion channel of the fan-out
MOVE_SCENE=$(line_of 'Arc::new(scene)')                # 248: `scene` moves into scene_ptr
MOVE_TABLE=$(line_of 'Arc::new(random_samples)')       # 271: `random_samples` moves into random_samples_ptr
FAN0=$(line_of 'for i in 0\.\.num_cpus')               # 275: first line of the fan-out
RECV=$(line_of 'rx\.recv()')                           # 302: inside the gather loop, which closes on the next line
FAN1=$((RECV + 1))                                     # 303
[ "$CHAN $MOVE_SCENE $MOVE_TABLE $FAN0 $FAN1" = "250 248 271 275 303" ] ||
    die "src/main.rs is not the revision this script was written against (anchors at $CHAN $MOVE_SCENE $MOVE_TABLE $FAN0 $FAN1)"
sed -n "${FAN1}p" "$M" | grep -q '^ *} *$' || die "line $FAN1 does not close the gather loop"

cp "$HERE/rtx_ffi.rs" "$HERE/render_gpu.rs" "$REF/src/"
cp "$HERE/build.rs" "$REF/build.rs"
grep -q '^build' "$REF/Cargo.toml" || sed -i '/^\[package\]/a build = "build.rs"' "$REF/Cargo.toml"

# 1. render(): the fan-out and its gather loop (spawn / render_pixel / put_pixel / send / recv) become one call; the
#    image is filled from the returned rows.  By then `scene` and `random_samples` have moved into their Arcs, so the
#    inserted code reads them through `scene_ptr` / `random_samples_ptr`.  `flat` is made in main() (step 3) and travels
#    in a global the size of a pointer.  The channel line goes too: with its only users deleted its type could not be
#    inferred.  Bottom-up, so the line numbers above stay valid.
sed -i "${FAN0},${FAN1}d" "$M"
sed -i "$((FAN0 - 1))r /dev/stdin" "$M" <<'RS'
    let table: Vec<rtx_ffi::Sample> = random_samples_ptr.iter().map(|s| rtx_ffi::Sample { s0: s.0, s1: s.1 }).collect();
    let flat = unsafe { &*FLAT.expect("main() flattens the primitives first") };
    let (rgb, stats) = render_gpu::render_frame(w, h, &scene_ptr.camera, &scene_ptr.light.primitives[0], flat, &table,
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
sed -i "${CHAN}d" "$M"

# 2. module declarations and the global, in front of the crate's own first `mod` line at the top of the file
sed -i '0,/^mod /s//mod rtx_ffi;\nmod render_gpu;\nstatic mut FLAT: Option<*const render_gpu::FlatScene> = None;\nmod /' "$M"

# 3. main(): flatten the Vec<Primitive> in front of the statement that builds the Scene — BoundingVolumeHierarchy::new,
#    INSIDE that struct literal, takes the Vec by value, so the statement as a whole is what the lines must precede.
SCENE=$(line_of '^ *let scene = Scene *{')
BVH=$(line_of 'BoundingVolumeHierarchy::new(primitives)')
[ "$SCENE" -lt "$BVH" ] || die "the Scene literal does not hold the BoundingVolumeHierarchy::new call"
sed -i "$((SCENE - 1))r /dev/stdin" "$M" <<'RS'
    let flat = Box::new(render_gpu::flatten(&primitives));
    unsafe { FLAT = Some(Box::into_raw(flat) as *const render_gpu::FlatScene); }
RS
echo "patched $M; build with: RTX_LIB_DIR=$LIB cargo build --release"
