"""One rank of a multi-process run of the distributed pipeline (started by tests/test_gpu_distributed.py, several ranks
on ONE GPU): gloo carries the library's all-to-all (Comm.torch), every rank saves the result it holds."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def result_arrays(p):
    """The state a pipeline holds after pre_process as flat arrays (contig strings, member lists, id lists)."""
    cs = p.contigs()
    out = {"ref": np.frombuffer(b"".join(r for r, _ in cs), dtype=np.uint8),
           "ref_len": np.array([len(r) for r, _ in cs], dtype=np.int64),
           "mem": np.concatenate([m for _, m in cs]) if cs else np.zeros(0, np.uint64),
           "mem_len": np.array([len(m) for _, m in cs], dtype=np.int64)}
    for name in ("sg", "allA", "allT", "allN", "fpA", "fpT", "fpN", "Nfile"):
        out[name] = p.id_list(name)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int); ap.add_argument("--world", type=int); ap.add_argument("--port", type=int)
    ap.add_argument("--reads"); ap.add_argument("--out"); ap.add_argument("--bounds", default="")
    ap.add_argument("--params", default="{}"); ap.add_argument("--dump", default="")
    ap.add_argument("--transport", default="gloo", help="gloo: the library's all-to-all through torch.distributed, every rank on GPU 0; rccl: ncclSend / ncclRecv, rank r on GPU r")
    a = ap.parse_args()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(a.port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)
    from minicom_amd.distributed import Comm, DistPipeline
    reads = np.load(a.reads)
    n, L = reads.shape
    bounds = [int(x) for x in a.bounds.split(",")] if a.bounds else [n * q // a.world for q in range(a.world + 1)]
    lo, hi = bounds[a.rank], bounds[a.rank + 1]
    device = 0
    if a.transport == "rccl":                                   # the production transport between real peers: one GPU per rank
        import torch
        device = a.rank
        torch.cuda.set_device(device)
        box = [Comm.unique_id() if a.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        comm = Comm.rccl(a.rank, a.world, box[0], device)
    else:
        comm = Comm.torch()
    p = DistPipeline(reads[lo:hi], lo, n, comm, L=L, device=device, host_threads=2, **json.loads(a.params))
    p.pre_process()
    res = result_arrays(p)
    res["stats"] = np.array([p.stat("rounds"), p.stat("merge_rounds"), p.stat("passes"), p.stat("big_bins"), p.stat("x_records")])
    np.savez(os.path.join(a.out, f"rank{a.rank}.npz"), **res)
    if a.dump and a.rank == a.world - 1:                      # any rank can write the archive: the last one does, for a change
        d = os.path.join(a.out, "streams"); os.makedirs(d)
        p.cluster_dump(d, order=a.dump == "order", paired=a.dump == "pe")
    p.close()
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
