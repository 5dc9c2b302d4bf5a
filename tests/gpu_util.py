"""Helpers for the GPU parity tests: build device-resident batches from host payloads."""
import importlib

import numpy as np

mz = importlib.import_module("minizip-ng_amd")


def make_batch(payloads, out_caps, align=16, device="cuda:0"):
    """payloads: list[bytes] of raw-deflate streams; out_caps: list[int].
    -> dict of CUDA tensors laid out the way the C ABI wants them."""
    import torch

    n = len(payloads)
    in_len = np.array([len(p) for p in payloads], dtype=np.int64)
    in_off = np.zeros(n, dtype=np.int64)
    pos = 0
    for i in range(n):
        in_off[i] = pos
        pos += (int(in_len[i]) + align - 1) // align * align
    blob = np.zeros(max(pos, 16), dtype=np.uint8)
    for i, p in enumerate(payloads):
        if p:
            blob[in_off[i]:in_off[i] + len(p)] = np.frombuffer(p, dtype=np.uint8)
    out_cap = np.array(out_caps, dtype=np.int64)
    out_off = np.zeros(n, dtype=np.int64)
    pos = 0
    for i in range(n):
        out_off[i] = pos
        pos += (int(out_cap[i]) + align - 1) // align * align
    dev = torch.device(device)
    return dict(
        d_in=torch.from_numpy(blob).to(dev), in_off=torch.from_numpy(in_off).to(dev),
        in_len=torch.from_numpy(in_len.astype(np.int32)).to(dev),
        d_out=torch.zeros(max(pos, 16), dtype=torch.uint8, device=dev), out_off=torch.from_numpy(out_off).to(dev),
        out_cap=torch.from_numpy(out_cap.astype(np.int32)).to(dev), h_out_off=out_off, n=n)


def run_inflate(batch):
    import torch

    out_len, in_used, crc, status = mz.inflate_batch(batch["d_in"], batch["in_off"], batch["in_len"], batch["d_out"],
                                                     batch["out_off"], batch["out_cap"])
    torch.cuda.synchronize()
    return (out_len.cpu().numpy().astype(np.int64), in_used.cpu().numpy().astype(np.int64), mz.u32(crc),
            status.cpu().numpy())


def entry_bytes(batch, h_out, i, n):
    o = int(batch["h_out_off"][i])
    return h_out[o:o + n].tobytes()
