#!/usr/bin/env python3
"""Probe of the packed-fp32 write-after-read (csrc/api.hip: pk_war_probe_kernel): the probe kernel on one stream, matrix-core GEMMs
(hipBLASLt, bf16) on two more; counts lanes whose accumulators are not the exact expected sums.  Usage: probe_pk_war.py [corun: mm|none]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from effi_mvs_plus_amd import _lib  # noqa: E402

dev = "cuda:0"
corun = sys.argv[1] if len(sys.argv) > 1 else "mm"
L = _lib.lib()
blocks, reps = 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n = blocks * 256
tid = torch.arange(n, device=dev)
x = 1.0 + (tid & 15).float() * 0.0625
# 16 * reps exact additions of x (4 fractional bits: sums stay exact in fp32) and of 2
want0, want1 = x * (16 * reps), torch.full((n,), 2.0 * 16 * reps, device=dev)
g = torch.Generator().manual_seed(0)
A = torch.randn(2048, 2048, generator=g).to(dev).bfloat16()
B = torch.randn(2048, 2048, generator=g).to(dev).bfloat16()
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
for variant in (0, 1, 3):
    bad_runs = bad_lanes = 0
    lanes_hist = torch.zeros(64, dtype=torch.long, device=dev)
    for it in range(40):
        outs = []
        for rep in range(6):
            if corun == "mm":
                with torch.cuda.stream(s1):
                    A @ B
                with torch.cuda.stream(s2):
                    B @ A
            with torch.cuda.stream(s0):
                o = torch.empty(2 * n, device=dev)
                rc = L.effi_debug_pk_war_probe(o.data_ptr(), blocks, reps, variant, torch.cuda.current_stream().cuda_stream)
                assert rc == 0, rc
                outs.append(o)
        torch.cuda.synchronize()
        for o in outs:
            w0 = (o[0::2] != want0) | (o[1::2] != want1)
            if w0.any():
                bad_runs += 1
                bad_lanes += int(w0.sum())
                lanes_hist += torch.bincount(tid[w0] & 63, minlength=64)
    hist = lanes_hist.cpu().tolist()
    print(f"variant {variant} ({['pk_fma then v_mov on its source', 'pk_fma, s_nop, v_mov', '', 'two v_fma then v_mov'][variant]}), co-runner {corun}: "
          f"launches with wrong lanes {bad_runs} of 240, wrong lanes {bad_lanes}; by lane quarter {[sum(hist[q * 16:(q + 1) * 16]) for q in range(4)]}")
