"""The reference's `examples/png` (examples/png/main.rs:43-61) on this library: Renderer::new((1600, 1200)) -> load_gltf ->
render_to_host_memory (16 warm-up frames) -> PNG.

    python examples/png.py path/to/scene.glb [out.png] [--size 1600x1200] [--camera px,py,pz,tx,ty,tz,fov]

With the reference checked out next to this repo: python examples/png.py /root/reference/examples/assets/ReflectionRoom.glb
(the example's camera is the default below). Needs a GPU: the product path has no CPU fallback.
"""
import argparse
import os
import struct
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_png(path, rgba):
    h, w, _ = rgba.shape
    raw = b"".join(b"\x00" + rgba[y].tobytes() for y in range(h))

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("gltf")
    ap.add_argument("out", nargs="?", default="render.png")
    ap.add_argument("--size", default="1600x1200")
    ap.add_argument("--camera", default="13,30,25,0,13,0,45", help="position, target, fov_y in degrees (main.rs:52-55)")
    args = ap.parse_args()
    from sunray_amd import runtime as rt
    w, h = (int(v) for v in args.size.split("x"))
    c = [float(v) for v in args.camera.split(",")]
    r = rt.Renderer((w, h))
    _group, instances = r.load_gltf(args.gltf)
    image = r.render_to_host_memory((tuple(c[0:3]), tuple(c[3:6]), c[6]), instances)
    write_png(args.out, image)
    print("You can find your render here: %s" % args.out)


if __name__ == "__main__":
    main()
