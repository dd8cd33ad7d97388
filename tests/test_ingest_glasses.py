"""Glass names -> indices in rayoptics_amd.ingest (the default must
not silently trace an achromatic n = 1.5 system).

The dispersion formulas on file are checked against each glass's catalogue
(nd, vd); the default lookup is dispersive, warns about and records what it
could not resolve; the reference-importer configuration (every named glass
n = 1.5, rayoptics/seq/medium.py:172-203 with an empty catalogue) has to be
asked for by name."""
import os
import warnings

import numpy as np
import pytest

import rayoptics_amd  # noqa: F401
from rayoptics_amd import ingest

REF = '/root/reference/src/rayoptics'


def test_every_formula_reproduces_its_catalogue_nd_and_vd():
    assert set(ingest.ND_VD) == set(ingest.SELLMEIER)
    for name, (nd, vd) in ingest.ND_VD.items():
        d = ingest.sellmeier_index(name, 587.5618)
        v = (d - 1.0) / (ingest.sellmeier_index(name, 486.1327) - ingest.sellmeier_index(name, 656.2725))
        assert abs(d - nd) < 2e-5, (name, d, nd)
        assert abs(v - vd) < 0.03, (name, v, vd)


def test_name_spellings():
    for spelled, canon in (('nbk7_schott', 'N-BK7'), ('NSK16_SCHOTT', 'N-SK16'), ('N-BK7', 'N-BK7'),
                           ('bk7', 'N-BK7'), ('f2', 'F2'), ('NSF5_SCHOTT', 'N-SF5'),
                           ('F5_SCHOTT', 'F5'), ('SILICA', 'SILICA')):
        assert ingest._canon(spelled) == canon
        assert ingest.knows_glass(spelled)
    assert not ingest.knows_glass('S-FTM16')


def test_model_glass_dispersion_goes_through_nd_and_vd():
    nd, vd = 1.5168, 64.17
    assert ingest.model_glass_index(nd, vd, 587.5618) == pytest.approx(nd, abs=1e-15)
    nF, nC = ingest.model_glass_index(nd, vd, 486.1327), ingest.model_glass_index(nd, vd, 656.2725)
    assert (nd - 1) / (nF - nC) == pytest.approx(vd, rel=1e-12)
    assert ingest.model_glass_index(nd, 0.0, 450.0) == nd


def _prescription(glasses):
    p = ingest.Prescription()
    for k in range(len(glasses) + 2):
        s = ingest.Ifc()
        s.cv = 0.01 if 0 < k <= len(glasses) else 0.0
        s.max_aperture = 10.0
        p.ifcs.append(s)
    p.thi = [1e10] + [3.0] * len(glasses)
    p.media = [('air',)] + [('glass', g) for g in glasses]
    p.wvls = [486.1327, 587.5618, 656.2725]
    p.ref_wvl = 1
    p.stop = 1
    return p


def test_default_lookup_is_dispersive_and_silent_for_known_glasses():
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        tbl = _prescription(['N-BK7', 'nsf5_schott']).to_table()
    assert tbl.fallback_glasses == ()
    n = tbl.n_table
    assert n[1, 1] == pytest.approx(1.5168, abs=2e-5) and n[1, 2] == pytest.approx(1.67271, abs=2e-5)
    assert n[0, 1] > n[1, 1] > n[2, 1] and n[0, 2] > n[1, 2] > n[2, 2]


def test_default_lookup_warns_about_and_records_unknown_glasses():
    with pytest.warns(ingest.UnknownGlassWarning, match='S-FTM16'):
        tbl = _prescription(['N-BK7', 'S-FTM16', 'E-LAFH3', 'S-FTM16']).to_table()
    assert tbl.fallback_glasses == ('S-FTM16', 'E-LAFH3')
    assert (tbl.n_table[:, 2] == 1.5).all() and (tbl.n_table[:, 3] == 1.5).all()
    assert tbl.n_table[0, 1] != tbl.n_table[2, 1]


def test_reference_importer_configuration_has_to_be_asked_for():
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        tbl = _prescription(['N-BK7', 'S-FTM16']).to_table(index_of=ingest.reference_fallback_index)
    assert (tbl.n_table[:, 1:3] == 1.5).all()
    assert tbl.fallback_glasses == ()


def test_custom_lookup_is_used_as_given():
    seen = []

    def lookup(name, wvl):
        seen.append(name)
        return 1.7 + 1e-5 * (600.0 - wvl)
    tbl = _prescription(['WHATEVER']).to_table(index_of=lookup)
    assert seen == ['WHATEVER'] * 3 and tbl.n_table[1, 1] == 1.7 + 1e-5 * (600.0 - 587.5618)


@pytest.mark.skipif(not os.path.isdir(REF), reason='reference tree not present')
def test_double_gauss_seq_default_indices_match_the_codev_listing():
    """rayoptics/codev/tests/ag_dblgauss.lis:30-34 lists the indices CODE V used"""
    pres = ingest.read_seq(os.path.join(REF, 'codev/tests/ag_dblgauss.seq'))
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        tbl = pres.to_table()
    assert tbl.fallback_glasses == ()
    glass_cols = [i for i, m in enumerate(pres.media) if m[0] == 'glass']
    assert glass_cols
    n = tbl.n_table[:, glass_cols]
    assert ((n > 1.55) & (n < 1.7)).all()
    assert (np.diff(n[np.argsort(tbl.wvls)], axis=0) < 0).all()     # normal dispersion


def test_every_glass_alias_resolves_to_dispersion_data():
    """'F_SILICA' contains the character the catalogue suffix is split at and never
    reached its alias (it fell back to n = 1.5 with a warning).  Every alias, with and without
    a catalogue suffix, in either case, reaches the Sellmeier data of its target"""
    from rayoptics_amd import ingest
    assert ingest.ALIASES
    for alias, target in ingest.ALIASES.items():
        assert target in ingest.SELLMEIER
        want = ingest.sellmeier_index(target, 587.6)
        forms = [alias, alias.lower()]
        if '_' not in alias:
            forms.append(alias + '_SCHOTT')
        for name in forms:
            assert ingest.knows_glass(name), name
            assert ingest.nominal_index(name, 587.6) == want, name
    assert abs(ingest.nominal_index('F_SILICA', 587.6) - 1.4585) < 2e-4
    assert abs(ingest.nominal_index('BK7', 587.6) - 1.5168) < 1e-4
