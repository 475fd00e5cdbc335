"""Build-container-only check of INTEGRATION.md section 1: the HIP mixins placed IN FRONT OF the real reference
classes.  `/root/reference` does not exist on the GPU box, so the whole module is skipped there; no GPU call is
made -- this pins the Python-level contract (method resolution order, name mangling, signatures, the attributes the
reference constructors set and the mixins read)."""
import inspect
import os
import sys
import types

import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")


REF_MODULES = ("ba_processor", "campose_processor", "epipolar_processor", "triangulation_processor", "utils",
               "view_processor", "key_tracker")


@pytest.fixture(scope="module")
def ref():
    """The reference's modules, imported for this test module only: sys.path, sys.modules (the reference's generically
    named `utils`, the inert `cv2` placeholder) and sys.dont_write_bytecode are restored on teardown so that no later
    test of the session picks them up."""
    saved_path, saved_flag = list(sys.path), sys.dont_write_bytecode
    saved_mods = {name: sys.modules.get(name) for name in REF_MODULES + ("cv2",)}
    sys.dont_write_bytecode = True                    # the reference directory is read-only
    sys.path.insert(0, REF)
    if "cv2" not in sys.modules:
        sys.modules["cv2"] = types.ModuleType("cv2")  # inert placeholder: OpenCV is absent, the hot path never calls it
    for name in REF_MODULES:
        sys.modules.pop(name, None)
    import ba_processor
    import campose_processor
    import epipolar_processor
    import triangulation_processor
    import utils
    yield types.SimpleNamespace(ba=ba_processor, cam=campose_processor, epi=epipolar_processor,
                                tri=triangulation_processor, utils=utils)
    sys.path[:] = saved_path
    sys.dont_write_bytecode = saved_flag
    for name, mod in saved_mods.items():
        if mod is None:
            sys.modules.pop(name, None)
        else:
            sys.modules[name] = mod


def _subclasses(sfm, ref):
    hip = sfm.processors

    class TriangulationProcessor(hip.HipTriangulationMixin, ref.tri.TriangulationProcessor):
        pass

    class CamposeProcessor(hip.HipCamposeMixin, ref.cam.CamposeProcessor):
        pass

    class BaProcessor(hip.HipBaMixin, ref.ba.BaProcessor):
        pass

    class EpipolarProcessor(hip.HipEpipolarMixin, ref.epi.EpipolarProcessor):
        pass

    return TriangulationProcessor, CamposeProcessor, BaProcessor, EpipolarProcessor


def test_mixins_override_the_hot_path_of_the_real_classes(sfm, ref):
    hip = sfm.processors
    tri_cls, cam_cls, ba_cls, epi_cls = _subclasses(sfm, ref)
    overrides = [
        (tri_cls, hip.HipTriangulationMixin, ref.tri.TriangulationProcessor,
         ["nonlinear_triangulate", "construct_jacobian_matrix", "triangulate", "linear_triangulate"]),
        (cam_cls, hip.HipCamposeMixin, ref.cam.CamposeProcessor,
         ["nonlinear_estimate_cam_pose_pnp", "construct_jacobian_matrix", "linear_estimate_cam_pose_pnp",
          "estimate_cam_pose_pnp", "extract_cam_pose_from_essential_mat", "evalulate_cam_pose_cheirality",
          "disambiguate_cam_pose_four"]),
        (epi_cls, hip.HipEpipolarMixin, ref.epi.EpipolarProcessor, ["determine_fundamental_mat", "extract_essential_mat"]),
    ]
    for cls, mixin, base, names in overrides:
        for name in names:
            assert getattr(cls, name) is getattr(mixin, name), (cls.__name__, name)       # the MRO picks the mixin
            assert hasattr(base, name), "the reference has no %s.%s to override" % (base.__name__, name)
            ours = list(inspect.signature(getattr(mixin, name)).parameters)
            theirs = list(inspect.signature(getattr(base, name)).parameters)
            assert ours == theirs, (name, ours, theirs)                                  # same argument names and order
    # the BA entry is a name-mangled private method: `process` calls self.__execute_bundle_adjustment(), which the
    # compiler spells _BaProcessor__execute_bundle_adjustment inside class BaProcessor (ba_processor.py:267)
    mangled = "_BaProcessor__execute_bundle_adjustment"
    assert mangled in ref.ba.BaProcessor.process.__code__.co_names
    assert mangled in vars(ref.ba.BaProcessor)
    assert getattr(ba_cls, mangled) is getattr(hip.HipBaMixin, mangled)
    assert getattr(ba_cls, mangled) is hip.HipBaMixin.execute_bundle_adjustment
    # everything else stays the reference's
    assert ba_cls.process is ref.ba.BaProcessor.process
    assert tri_cls.add_tri_pt is ref.tri.TriangulationProcessor.add_tri_pt


def test_reference_constructors_provide_what_the_mixins_read(sfm, ref, capsys):
    tri_cls, cam_cls, ba_cls, epi_cls = _subclasses(sfm, ref)
    ransac = ref.utils.RansacConfig(8.0, 0.99, 0.75, 6, 300)                      # ba_processor.py:463-480
    tp = tri_cls()                                                                # ba_processor.py:485
    cp = cam_cls(ransac, 5, 300)                                                  # ba_processor.py:486
    ep = epi_cls(ransac)                                                          # ba_processor.py:484
    bp = ba_cls(None, None, ep, tp, cp)                                           # ba_processor.py:487-488
    capsys.readouterr()
    assert (tp.damping_factor, tp.iteration) == (0.5, 100)                        # tri:12 defaults the mixin falls back to
    assert (cp.damping_factor, cp.iteration) == (5, 300) and cp.ransac_config is ransac
    assert ep.ransac is ransac and ep.fund_mat.shape == (3, 3)
    assert (bp.iteration, bp.damping_factor) == (3, 5)                            # ba:24 defaults
    for attr in ("view_processor", "key_tracker", "tri_processor", "campose_processor"):
        assert hasattr(bp, attr)
    assert bp.tri_processor is tp
    # the drop-in's own RansacConfig mirrors the reference's field for field
    ours = sfm.processors.RansacConfig(8.0, 0.99, 0.75, 6, 300)
    capsys.readouterr()
    for field in ("inlier_threshold", "subset_confidence", "sample_confidence", "sample_num", "iteration", "random_seed"):
        assert getattr(ours, field) == getattr(ransac, field), field
    # class-level switches of the mixins do not collide with reference attributes
    for name in ("ba_quirk_flags", "ba_verbose", "ba_resident", "ba_last_action"):
        assert not hasattr(ref.ba.BaProcessor, name)
    assert not hasattr(ref.cam.CamposeProcessor, "quirk_flags")
