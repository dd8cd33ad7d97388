#!/usr/bin/env python3
"""cProfile of SpotDiagramFigure.update_data() through the drop-ins at a large num_rays (GPU box,
staged reference)."""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))


def main(model='dblgauss', num=256):
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    import refmodels as ref
    import rayoptics_amd  # noqa: F401
    from rayoptics_amd import install
    from rayoptics.mpl.axisarrayfigure import SpotDiagramFigure
    opm = getattr(ref, model)()
    install.install()

    def run():
        fig = plt.figure(FigureClass=SpotDiagramFigure, opt_model=opm, num_rays=num)
        fig.update_data()
        plt.close(fig)
    for _ in range(3):
        run()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        run()
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(28)
    print(out.getvalue()[:6500])
    install.uninstall()


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'dblgauss', int(sys.argv[2]) if len(sys.argv) > 2 else 256)
