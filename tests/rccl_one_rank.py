"""Child process of tests/test_gpu_rccl.py: the tile-parallel frame loop of bench.py --gpus N with ONE rank over RCCL.

init_process_group("nccl", world_size = 1) -> FramePipeline (two frames in flight) + GatherPipeline (asynchronous
all_gather_into_tensor on RCCL's stream, ordered after the final pass's stream) for a few frames -> the gathered image must
equal the frame buffer bit for bit and the frames of a plain sequential loop -> destroy_process_group. Prints
"RCCL_ONE_RANK_OK <crc32>" on success. Run as its own process: a process group is per process."""
import os
import socket
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    import numpy as np
    import torch
    import torch.distributed as dist
    from sunray_amd import abi, distributed as sd, runtime as rt, scenes
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda:0"))
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    W, H = 328, 184                       # ragged: not a multiple of the 8x8 tile
    desc = scenes.heightfield(n=200)
    bn = scenes.white_noise_rgba8()
    scene = rt.Scene(0).load(desc)
    cfg = abi.SrTraceConfig.reference()
    part = sd.Partition(W, H, 1, "cols")
    pipe = sd.GatherPipeline(part, 0, "cuda:0")
    fpipe = sd.FramePipeline(rt.DeviceFrame(W, H, bn), rt.DeviceFrame(W, H, bn))
    seq = rt.DeviceFrame(W, H, bn)       # the same frames, one after the other, no gather
    prev = None
    crcs = []
    for f in range(frames):
        m = rt.camera_matrices(desc.camera_pos, desc.camera_target, desc.fov_y, W, H, prev)
        prev = list(m.view_proj)
        fr = fpipe.step(scene, m, f, cfg, part, 0, after_final=lambda x: pipe.submit(x.raw_color))
        assert pipe.work[pipe.last] is not None, "the collective was bypassed"
        sd.render_strip(scene, seq, m, f, cfg, part, 0)
        img = pipe.image()               # waits for this frame's collective on the current stream
        torch.cuda.synchronize()
        a, b, c = img.cpu().numpy(), fr.raw_color.cpu().numpy(), seq.raw_color.cpu().numpy()
        assert a.shape == (W * H, 4) and np.array_equal(a.view(np.uint32), b.view(np.uint32)), "gathered image != frame buffer (frame %d)" % f
        assert np.array_equal(a.view(np.uint32), c.view(np.uint32)), "pipelined + gathered frame != sequential frame (frame %d)" % f
        assert np.isfinite(a).all() and a[:, :3].any()
        crcs.append(zlib.crc32(a.tobytes()) & 0xFFFFFFFF)
    # a reduction over the same group, as bench.py does for its timing / ray totals
    t = torch.tensor([3.5], dtype=torch.float64, device="cuda:0")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 3.5
    pipe.wait()
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_ONE_RANK_OK %08x" % crcs[-1], flush=True)


if __name__ == "__main__":
    main()
