"""The dominant dense kernels alone, for counter passes: the 3x3 forward / data gradient (conv3x3_k32_nhwc_bf16_kernel<128,4,3,false>) and the
weight gradient (conv3x3_wgrad_kernel<128,128,3,1,3>) at the benchmark's shape 128 -> 128 @ 4 x 188 x 188, N launches each.
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python tools/conv_pmc_bench.py [N]
    python tools/pmc_kernel.py conv3x3 <dir> ...      # per-launch means"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    from sparse2dense_amd import dense2d as D
    dev = "cuda:0"
    torch.manual_seed(0)
    w = torch.randn(128, 128, 3, 3, device=dev) * 0.05
    x = torch.randn(4, 128, 188, 188, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(4, 128, 188, 188, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    packed = D.pack_weights(w)
    for _ in range(n):
        D.conv3x3_nhwc(x, packed, None, 128, 128, 1)
        D.conv3x3_wgrad(x, dy, 1)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
