"""Model -> layout -> validate: product (C++) vs oracle vs the reference README's printed layouts."""
import numpy as np
import pytest

from helpers import golden, make_grid, platform_defs
from oracle import encoder_oracle as eo, layout_oracle as lo
from timberborn_support_solver_amd import PLATFORMS_DEFAULT, Encoding, PlatformLayout, WorldGrid

README = golden("readme_layouts.json")["layouts"]


@pytest.mark.parametrize("lay", README, ids=lambda l: f"{l['marked']}-supports")
def test_readme_layouts_validate_with_distance_4(lay):
    grid = WorldGrid.from_rows(lay["terrain_rows"])
    plats = [(x, y, 1, 1, 0) for x, y in lay["supports_xy"]]
    p = PlatformLayout.from_platforms(plats)
    assert p.platform_count() == lay["marked"]
    assert p.validate(grid).is_valid()
    res = lo.validate({(x, y): ((1, 1), False) for x, y in lay["supports_xy"]}, eo.grid_from_rows(lay["terrain_rows"]))
    assert lo.is_valid(res)
    # removing any support leaves unsupported terrain in at least one of the layouts' neighbourhoods,
    # and the validator must notice a support moved off the grid / onto another one
    broken = PlatformLayout.from_platforms(plats[1:] + [(plats[0][0] + 40, plats[0][1], 1, 1, 0)])
    assert broken.validate(grid).n_out_of_bounds_platforms == 1


def test_validator_flags_overlap_oob_and_unsupported():
    grid = make_grid("rect8x8")
    v = PlatformLayout.from_platforms([(0, 0, 3, 3, 0), (2, 2, 1, 1, 0)]).validate(grid)
    assert v.n_overlapping_platforms == 2 and v.n_unsupported_terrain > 0
    v = PlatformLayout.from_platforms([(6, 6, 5, 5, 0)]).validate(grid)
    assert v.n_out_of_bounds_platforms == 1
    # rotated 1x6: occupies 6x1
    v = PlatformLayout.from_platforms([(1, 3, 1, 6, 1), (1, 4, 1, 1, 0)]).validate(grid)
    assert v.n_overlapping_platforms == 0 and v.n_out_of_bounds_platforms == 0
    o = lo.validate({(1, 3): ((1, 6), True), (1, 4): ((1, 1), False)}, eo.grid_rect(8, 8))
    assert len(o[0]) == v.n_unsupported_terrain


@pytest.mark.parametrize("terrain,pset", [("ex1", "default"), ("ex3", "default"), ("rect8x8", "default"), ("rect8x8", "1x1")])
def test_from_assignment_and_validate_match_oracle_on_random_models(terrain, pset):
    grid = make_grid(terrain)
    enc = Encoding.encode(platform_defs(pset), grid)
    o = eo.Encoding(platform_defs(pset), eo.grid_from_rows(grid.rows()))
    rng = np.random.default_rng(7)
    for density in (0.0, 0.01, 0.05, 0.3):
        model = np.where(rng.random(enc.n_vars) < density, 1, -1).astype(np.int8)
        p = PlatformLayout.from_assignment(model, enc)
        q = lo.from_assignment(model.tolist(), o)
        assert sorted(p.platforms()) == sorted((x, y, d[0], d[1], int(r)) for (x, y), (d, r) in q.items())
        v, r = p.validate(grid), lo.validate(q, o.grid)
        assert (v.n_unsupported_terrain, v.n_overlapping_platforms, v.n_out_of_bounds_platforms) == tuple(len(s) for s in r)
