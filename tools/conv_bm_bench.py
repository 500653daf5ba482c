"""Forward time of the dense 3x3 tile kernel per tile height (S2D_CONV_BM = 128 / 96 / 64 forces it; unset = the library's choice) at the
benchmark's layer shapes.   S2D_CONV_BM=96 python tools/conv_bm_bench.py
S2D_CONV_WS=1: the warp-specialised form (r06 experiment, csrc/conv2d_nhwc.hip conv3x3_k32ws_nhwc_bf16_kernel); the checksum column must not move."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    from sparse2dense_amd import _lib, dense2d as D
    dev = "cuda:0"
    lib = _lib.load()
    for (cin, cout, h) in [(128, 128, 188), (256, 256, 94), (256, 128, 188), (128, 256, 188), (256, 512, 94), (512, 64, 188), (64, 64, 188)]:
        torch.manual_seed(0)
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
        x = torch.randn(4, cin, h, h, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        packed = D.pack_weights(w)
        for _ in range(3):
            D.conv3x3_nhwc(x, packed, None, cin, cout, 1)
        torch.cuda.synchronize()
        n = 20
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            D.conv3x3_nhwc(x, packed, None, cin, cout, 1)
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / n * 1e3
        out = D.conv3x3_nhwc(x, packed, None, cin, cout, 1)
        chk = f"{float(out.double().sum()):.6e}/{float(out.double().abs().max()):.4e}"
        fl = 2.0 * 4 * h * h * 9 * cin * cout
        print(f"BM={os.environ.get('S2D_CONV_BM', 'auto'):>4s} {cin:3d}->{cout:3d} @4x{h}^2 rows/tile {lib.s2d_conv2d3x3_tile_rows(4, h, h, cin, cout, 1, 1)}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  ws={os.environ.get('S2D_CONV_WS', '0')} checksum {chk}", flush=True)


if __name__ == "__main__":
    main()
