"""tools/determinism_check.py <frames> [--scene stock|mixed|scaled] [--grid] — renders the same 1200x800x60 frame <frames> times
with the library named by RTK_LIB (or the in-tree one) and reports how many distinct images came out (the image is
defined by pixel, sample and seed alone: one).  --scene mixed: 200 spheres of varied radius (general words, the loop the
stock scene never enters); --scene scaled: the 10 001-sphere scene of BASELINE configs[3] at 8 spp (the word-local matrix-pipe
walk of streamed scenes).  --grid: one more frame through the uniform-grid kernel (no matrix instructions, another
code path to the same closest hits): it must equal the list walk's image bit for bit, which is what tells a
deterministically WRONG list walk from a right one."""
import hashlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracinginoneweekend_amd as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
scene = sys.argv[sys.argv.index("--scene") + 1] if "--scene" in sys.argv else "stock"
depth = int(sys.argv[sys.argv.index("--depth") + 1]) if "--depth" in sys.argv else 50      # --depth N: which parts of the kernel other waves run (hazard studies)
ctx = rt.Context(0)
ctx.set_scene(rt.stock_scene() if scene == "stock" else rt.scaled_scene(k=50) if scene == "scaled" else rt.mixed_radii_scene(200))
print("scene", scene, ctx.scene_layout(), flush=True)
W, H, spp = 1200, 800, (8 if scene == "scaled" else 60)
cam = rt.stock_camera(W / H)
seen = {}
for i in range(n):
    img, st = ctx.render(cam, W, H, spp, depth, seed=11)
    h = hashlib.sha256(img.tobytes()).hexdigest()[:12]
    h = h + ":" + str(st["exact_tests"]) if "--count-tests" in sys.argv else h      # --count-tests: a frame is (image, exact tests): for builds whose image is blank
    seen[h] = seen.get(h, 0) + 1
    print(i, h, st["rays"], st["exact_tests"], st.get("samples"), st.get("lane_slots"), flush=True)
ok = len(seen) == 1
if "--grid" in sys.argv:
    assert ctx.set_accel(rt.ACCEL_GRID) == rt.ACCEL_GRID
    img, st = ctx.render(cam, W, H, spp, depth, seed=11)
    h = hashlib.sha256(img.tobytes()).hexdigest()[:12]
    print("grid", h, st["rays"], flush=True)
    same = h in seen
    print("uniform-grid frame", "EQUALS" if same else "DIFFERS FROM", "the list walk's")
    ok = ok and same
print(os.environ.get("RTK_LIB", "in-tree"), "scene", scene, "distinct images:", len(seen))
sys.exit(0 if ok else 1)
