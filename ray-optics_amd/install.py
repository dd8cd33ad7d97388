"""Rebinds the reference's module-level seams to the
device-backed drop-ins.  Every caller in ray-optics reaches these functions
through a module-qualified name (``trace.trace_grid(...)``,
``seq_model.trace_grid(...)``), so rebinding the attribute is a true drop-in:
SpotDiagramFigure, RayFan, RayList and RayGrid consume the results unchanged.

    import rayoptics_amd.install as roxi
    roxi.install()              # unsupported models raise
    roxi.install('reference')   # unsupported models use the reference's own code
    roxi.uninstall()
"""
import functools

from . import session
from . import trace as _t
from . import analyses as _a
from . import vigcalc as _v
from .table import UnsupportedModelError

_saved = {}


def _guard(ours, theirs):
    @functools.wraps(theirs)
    def call(*args, **kwargs):
        try:
            return ours(*args, **kwargs)
        except UnsupportedModelError:
            if session.FALLBACK == 'reference':
                return theirs(*args, **kwargs)
            raise
    return call


def install(fallback='raise', tolerance_mode=None):
    import rayoptics.raytr.raytrace as rraytrace
    import rayoptics.raytr.trace as rtrace
    import rayoptics.raytr.analyses as ranalyses
    import rayoptics.raytr.opticalspec as ropticalspec
    import rayoptics.raytr.vigcalc as rvigcalc
    import rayoptics.raytr.wideangle as rwideangle
    from rayoptics.seq.sequential import SequentialModel
    if _saved:
        uninstall()
    session.FALLBACK = fallback
    if tolerance_mode is not None:      # session.TOLERANCE_MODE: ROX_FAST_FP64 on every launch
        session.set_tolerance_mode(tolerance_mode)
    seams = [(rraytrace, 'trace', _t.raytrace_trace),
             (rraytrace, 'trace_raw', _t.raytrace_trace_raw),
             (rtrace, 'trace_grid', _t.trace_grid), (rtrace, 'trace_fan', _t.trace_fan),
             (ranalyses, 'trace_list_of_rays', _a.trace_list_of_rays),
             (ranalyses, 'trace_ray_list', _a.trace_ray_list),
             (ranalyses, 'trace_ray_grid', _a.trace_ray_grid),
             (ranalyses, 'trace_ray_fan', _a.trace_ray_fan),
             (ranalyses, 'eval_fan', _a.eval_fan),
             (ranalyses, 'trace_fan', _a.trace_fan),
             (ranalyses, 'focus_fan', _a.focus_fan),
             (ranalyses, 'eval_wavefront', _a.eval_wavefront),
             (ranalyses, 'trace_wavefront', _a.trace_wavefront),
             (ranalyses, 'focus_wavefront', _a.focus_wavefront),
             (ranalyses, 'trace_pupil_coords', _a.trace_pupil_coords),
             (ranalyses, 'focus_pupil_coords', _a.focus_pupil_coords),
             (ranalyses, 'calc_psf', _a.calc_psf),
             (ranalyses, 'update_psf_data', _a.update_psf_data),
             (SequentialModel, 'trace_grid', _t.seq_trace_grid),
             (SequentialModel, 'trace_fan', _t.seq_trace_fan),
             (SequentialModel, 'trace_wavefront', _a.seq_trace_wavefront),
             # chief-ray aiming: trace.aim_chief_ray is imported by name into
             # opticalspec (opticalspec.py:19), so both bindings are replaced, and
             # update_optical_properties aims all fields in one launch
             # the reverse chief-ray iteration behind fields given as real image heights
             # (wideangle.eval_real_image_ht calls it module-qualified, wideangle.py:646)
             (rtrace, 'iterate_ray_raw', _t.iterate_ray_raw),
             # the five close rays per field point of trace_astigmatism_curve (AstigmatismCurvePlot)
             (rtrace, 'trace_astigmatism', _t.trace_astigmatism),
             # trace_field / trace_all_fields: a field's list of rays as one launch
             (rtrace, 'trace_ray_list_at_field', _t.trace_ray_list_at_field),
             # setup_pupil_coords -> get_chief_ray_pkg -> trace_chief_ray (bare module globals of
             # rayoptics.raytr.trace): the chief rays of all fields x wavelengths in one launch
             (rtrace, 'trace_chief_ray', _t.trace_chief_ray),
             (rtrace, 'trace_astigmatism_coddington_fan', _t.trace_astigmatism_coddington_fan),
             (rtrace, 'aim_chief_ray', _t.aim_chief_ray),
             (ropticalspec, 'aim_chief_ray', _t.aim_chief_ray),
             # the wide-angle pupil search behind aim_chief_ray (trace.py:634-635) and
             # eval_z_enp_curve; trace.py imports it by name (trace.py:29)
             (rwideangle, 'find_real_enp', _t.find_real_enp),
             (rtrace, 'find_real_enp', _t.find_real_enp),
             (ropticalspec.OpticalSpecs, 'update_optical_properties',
              _t.osp_update_optical_properties),
             # vignetting search and the boundary rays behind set_clear_apertures
             (rvigcalc, 'calc_vignetting_for_field', _v.calc_vignetting_for_field),
             (rvigcalc, 'set_vig', _v.set_vig),
             # vigcalc.set_pupil's marginal-ray iteration (set_pupil / set_stop_aperture call
             # iterate_pupil_ray, set_vig and set_clear_apertures through these module globals)
             (rvigcalc, 'iterate_pupil_ray', _v.iterate_pupil_ray),
             (rtrace, 'trace_boundary_rays_at_field', _v.trace_boundary_rays_at_field),
             # ... of every field in one launch (set_clear_apertures, every model update)
             (rtrace, 'trace_boundary_rays', _v.trace_boundary_rays)]
    for owner, name, ours in seams:
        theirs = getattr(owner, name)
        _saved[(owner, name)] = theirs
        setattr(owner, name, _guard(ours, theirs))


def uninstall():
    for (owner, name), theirs in _saved.items():
        setattr(owner, name, theirs)
    _saved.clear()
    session.clear()
