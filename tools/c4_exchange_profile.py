#!/usr/bin/env python3
"""cProfile of one pipelined pass of BASELINE configs[3] (5 fields x 256^2, one rank) through
both exchanges: where the host time of a SMALL sharded spot problem goes."""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import abi
    bench.SPOT_FLAGS = abi.INTERSECT_OBJ | abi.CHECK_APERTURES | abi.APPLY_VIGNETTING
    prob = bench.SpotProblem(torch, None, False, 1, 0, 'rc_telescope_c4', 256, 'field')
    fence = lambda: torch.cuda.synchronize()        # noqa: E731
    seg = prob.segment('prof', fence)
    for ex, kw in (('rccl', {}), ('host', {'segment': seg})):
        for _ in range(5):
            prob.run(ex, **kw)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            v = prob.run(ex, **kw)
            del v
        pr.disable()
        out = io.StringIO()
        pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(22)
        print('=====', ex)
        print(out.getvalue()[:6000])
    seg.close(unlink=True)
    prob.close()


if __name__ == '__main__':
    main()
