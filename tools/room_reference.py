#!/usr/bin/env python3
"""The one image the reference itself produced of a scene this build can make: renders/infinite_room.png is the `room`
scene (src/scene/scene.rs:445-573: red floor, glass sphere, mirror walls, emissive quad) seen from a camera the author
moved by hand, with depth of field switched on in the UI, after an unknown number of frames -- a window capture, 1429 x
799.  This script renders the same scene from a camera fitted to that picture (tools/room_reference.py --fit explains
how the fit was made) and exports it the way the reference does (src/core/app.rs:408-460: gamma 1 / 2.2, clamp, u8,
vertical flip) so that the two can be looked at side by side, and region statistics compared (tests/test_room_reference.py).

    python tools/room_reference.py --oracle 357 200 64 out.png     # CPU oracle (build container)
    python tools/room_reference.py --gpu 1429 799 512 out.png      # the product (GPU box)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ray_tracer_2_amd as rt  # noqa: E402

# The fitted camera.  The author's window shows the render texture unflipped: texel row 0 (the
# bottom of the view) at the TOP?  No: the blit draws v = 0 at the top of the quad and the picture is upright,
# so what the window shows is the image with rows reversed AND columns as stored; the export (app.rs:414,458-459) is the
# same orientation.  Position / direction / depth of field were fitted by hand against the picture's geometry: the two
# spheres' apparent sizes and positions fix the eye (sizes: 0.6 / 0.8 units across at 45 degrees vertical field of view),
# the blur of the far reflections fixes defocus_strength given focus_dist.
FIT = dict(origin=(2.02, 0.97, 1.24), look_at=(2.02 - 0.971, 0.97 + 0.025, 1.24 - 0.240), fov=52.0, focus_dist=2.4, defocus_strength=80.0,
           diverge_strength=0.0, bounces=12)


def camera_uniform(fit):
    """CameraUniform (src/scene/camera.rs:81-91) of an eye at `origin` looking at `look_at`: the matrix a free-fly camera
    ends up with -- right = up x forward, as in the Cornell camera's diag(-1, 1, -1).  (Transform::cam itself
    cannot be used for this: it stores glam's VIEW rotation as cam-to-world, which only coincides for yaws of 0 and 180
    degrees -- the scenes' own cameras.)"""
    from ray_tracer_2_amd import _abi as A
    o, la = np.asarray(fit["origin"], np.float64), np.asarray(fit["look_at"], np.float64)
    f = (la - o) / np.linalg.norm(la - o)
    r = np.cross([0.0, 1.0, 0.0], f)
    r /= np.linalg.norm(r)
    u = np.cross(f, r)
    cam = A.CameraUniform()
    for c, col in enumerate((r, u, f, o)):
        for k in range(3):
            cam.cam_to_world[c][k] = float(col[k])
        cam.cam_to_world[c][3] = 1.0 if c == 3 else 0.0
    fd = max(float(fit["focus_dist"]), 1.0)                       # camera.rs:75
    ph = fd * np.tan(np.radians(fit["fov"] * 0.5)) * 2.0          # camera.rs:83-84
    cam.view_params[0], cam.view_params[1], cam.view_params[2] = ph * 16.0 / 9.0, ph, fd
    cam.defocus_strength, cam.diverge_strength = float(fit["defocus_strength"]), float(fit["diverge_strength"])
    return cam


def room_arrays(fit=FIT, assets="/root/reference/assets"):
    arrays = rt.SceneArrays.from_scene(rt.Scene.from_name("room", assets))
    arrays.uniform.camera = camera_uniform(fit)
    # The picture's glass sphere is CLEAR (sharp refraction and reflections); scene.rs:556-562 leaves its smoothness at 0
    # and its specular at 0.1, which renders frosted (wgsl:432-433: the refracted direction is mixed with a diffuse one by
    # `smoothness`) -- the author turned the sliders of the material editor (src/rendering/egui.rs) before the capture.
    # The fit turns them too; everything else is the scene as committed.
    arrays.spheres["material"]["smoothness"][0] = fit.get("glass_smoothness", 1.0)
    arrays.spheres["material"]["specular"][0] = fit.get("glass_specular", 1.0)
    return arrays


def render(mode, w, h, frames, arrays, spp=4, bounces=5):
    if mode == "--oracle":
        from oracle import oracle
        img = np.zeros((h, w, 4), np.float32)
        for f in range(frames):
            img, _ = oracle.render(rt.make_params(w, h, bounces, spp, skybox=0, frames=f), arrays, image=img)
        return img, oracle.export_rgba8(img)
    tr = rt.RayTracer(0, w, h)
    tr.load_scene(arrays)
    tr.render_frames(rt.make_params(w, h, bounces, spp, skybox=0, frames=0), frames)
    img = tr.read_image(w, h)
    out = np.zeros((h, w, 4), np.uint8)
    img = np.ascontiguousarray(img, np.float32)
    assert tr._L.rt_export_rgba8(img.ctypes.data, w, h, out.ctypes.data) == 0   # (a host function: app.rs:408-460 on the frame read back)
    return img, out


def main():
    mode, w, h, frames, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    fit = dict(FIT)
    for kv in sys.argv[6:]:
        k, v = kv.split("=")
        fit[k] = tuple(float(x) for x in v.split(",")) if "," in v else float(v)
    fit["defocus_strength"] = fit["defocus_strength"] * w / 1920.0   # (wgsl:488: the lens jitter is strength / image WIDTH; the author's texture was 1920 wide)
    arrays = room_arrays(fit)
    img, rgba8 = render(mode, w, h, frames, arrays, bounces=int(fit.get("bounces", 5)))
    from PIL import Image
    Image.fromarray(rgba8[..., :3]).save(out)
    print("saved", out, "mean linear", img[..., :3].mean(axis=(0, 1)))


if __name__ == "__main__":
    main()
