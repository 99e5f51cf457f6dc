"""tools/determinism_check.py <frames> — renders the same frame <frames> times with the library named by RTK_LIB (or the
in-tree one) and reports how many distinct images came out (the image is defined by pixel, sample and seed alone: one)."""
import hashlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracinginoneweekend_amd as rt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = rt.Context(0)
ctx.set_scene(rt.stock_scene())
W, H, spp = 1200, 800, 60
cam = rt.stock_camera(W / H)
seen = {}
for i in range(n):
    img, st = ctx.render(cam, W, H, spp, 50, seed=11)
    h = hashlib.sha256(img.tobytes()).hexdigest()[:12]
    seen[h] = seen.get(h, 0) + 1
    print(i, h, st["rays"], st["exact_tests"], st.get("samples"), st.get("lane_slots"), flush=True)
print(os.environ.get("RTK_LIB", "in-tree"), "distinct images:", len(seen))
sys.exit(0 if len(seen) == 1 else 1)
