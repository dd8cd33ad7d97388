#!/usr/bin/env python3
"""cProfile of RayFanFigure.update_data() through the drop-ins on the GPU box (staged
reference): what the 12-14 ms of tools/figure_latency.py are made of."""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main():
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import refmodels as ref
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import install
    from rayoptics.mpl.axisarrayfigure import RayFanFigure
    opm = ref.dblgauss()
    install.install()

    def run():
        fig = plt.figure(FigureClass=RayFanFigure, opt_model=opm, data_type='Ray', num_rays=21)
        fig.update_data()
        plt.close(fig)
    for _ in range(3):
        run()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        run()
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(30)
    print(out.getvalue()[:7000])
    install.uninstall()


if __name__ == '__main__':
    main()
