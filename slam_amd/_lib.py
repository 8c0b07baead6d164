"""Loader for the in-tree HIP shared library (slam_amd/lib/libslam_amd.so).

The library is the product: there is no CPU fallback.  If it is missing, or a
call fails, this module raises -- it never routes around the GPU path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libslam_amd.so")

# Every symbol that include/slam_amd.h declares.
EXPORTS = (
    "slam_version",
    "slam_last_error",
    "slam_select_kernels",
    "slam_pack_descriptors",
    "slam_hamming_match",
    "slam_hamming_knn2",
    "slam_l2_match",
    "slam_l2_pack_int8",
    "slam_l2_match_int8",
    "slam_ratio_filter",
    "slam_row_filter",
    "slam_triangulate_dlt",
    "slam_triangulate_matched",
    "slam_join_consensus",
    "slam_join_consensus_prepared",
    "slam_score_hypotheses",
    "slam_score_hypotheses_range",
    "slam_score_hypotheses_static",
    "slam_select_best",
    "slam_build_inlier_tables",
    "slam_winner_tables",
    "slam_track_assign_work_bytes",
    "slam_track_assign",
    "slam_ransac_sample",
    "slam_p3p_hypotheses",
    "slam_ransac_select",
    "slam_refine_poses",
    "slam_refine_winners",
    "slam_refine_winners_stereo",
    "slam_ransac_adopt",
)


class SlamError(RuntimeError):
    """A slam_amd C-ABI call returned a negative status."""


_lib = None


def load():
    """dlopen the library once and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlamError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C slam_amd/csrc` (there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise SlamError(f"{LIB_PATH} does not export {name}")
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.slam_version.restype = ci
    lib.slam_version.argtypes = []
    lib.slam_last_error.restype = ctypes.c_char_p
    lib.slam_last_error.argtypes = []
    lib.slam_select_kernels.restype = ci
    lib.slam_select_kernels.argtypes = [ci, ci]
    lib.slam_pack_descriptors.restype = ci
    lib.slam_pack_descriptors.argtypes = [vp, ci, ci, ci, vp, ci, vp]
    lib.slam_hamming_match.restype = ci
    lib.slam_hamming_match.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, ci, cd, vp, vp, vp, vp, vp, vp]
    lib.slam_hamming_knn2.restype = ci
    lib.slam_hamming_knn2.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp, vp, vp]
    lib.slam_l2_match.restype = ci
    lib.slam_l2_match.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, cd, vp, vp, vp, vp, vp, vp]
    lib.slam_l2_pack_int8.restype = ci
    lib.slam_l2_pack_int8.argtypes = [vp, ci, ci, vp, vp, vp, vp]
    lib.slam_l2_match_int8.restype = ci
    lib.slam_l2_match_int8.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ci, ci, ci, cd, vp, vp, vp, vp, vp, vp]
    lib.slam_ratio_filter.restype = ci
    lib.slam_ratio_filter.argtypes = [vp, vp, ci, cd, vp, vp]
    lib.slam_row_filter.restype = ci
    lib.slam_row_filter.argtypes = [vp, vp, ci, cd, vp, vp]
    lib.slam_triangulate_dlt.restype = ci
    lib.slam_triangulate_dlt.argtypes = [vp, vp, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp]
    lib.slam_triangulate_matched.restype = ci
    lib.slam_triangulate_matched.argtypes = [vp, vp, vp, vp, vp, vp, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.slam_join_consensus.restype = ci
    lib.slam_join_consensus.argtypes = [vp, ci, ci] + [vp] * 11 + [ci] + [vp] * 9 + [vp]
    lib.slam_join_consensus_prepared.restype = ci
    lib.slam_join_consensus_prepared.argtypes = [vp, ci, ci] + [vp] * 11 + [ci] + [vp] * 9 + [vp, cd, vp, vp]
    lib.slam_score_hypotheses.restype = ci
    lib.slam_score_hypotheses.argtypes = [vp, ci, ci, vp, vp] + [vp] * 7 + [ci, vp, ci, cd, vp, vp, ci, vp, vp]
    lib.slam_score_hypotheses_range.restype = ci
    lib.slam_score_hypotheses_range.argtypes = ([vp, ci, ci, ci, ci, vp, vp, vp] + [vp] * 7 +
                                                [ci, vp, ci, cd, vp, vp, ci, vp, ci, vp])
    lib.slam_score_hypotheses_static.restype = ci
    lib.slam_score_hypotheses_static.argtypes = [vp, ci, ci, vp, vp] + [vp] * 7 + [ci, vp, ci, cd, vp, vp]
    lib.slam_select_best.restype = ci
    lib.slam_select_best.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    lib.slam_build_inlier_tables.restype = ci
    lib.slam_build_inlier_tables.argtypes = [vp, ci, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.slam_winner_tables.restype = ci
    lib.slam_winner_tables.argtypes = [vp, ci, vp, vp, ci, vp, vp, cd] + [vp] * 16
    lib.slam_track_assign_work_bytes.restype = ctypes.c_size_t
    lib.slam_track_assign_work_bytes.argtypes = [ci, ci, ci]
    lib.slam_track_assign.restype = ci
    lib.slam_track_assign.argtypes = [vp, vp, vp, ci, ci, ci, ci] + [vp] * 8 + [ctypes.c_size_t, vp]
    lib.slam_ransac_sample.restype = ci
    lib.slam_ransac_sample.argtypes = [ctypes.c_uint64, ci, ci, vp, ci, vp, vp]
    lib.slam_p3p_hypotheses.restype = ci
    lib.slam_p3p_hypotheses.argtypes = [vp] * 6 + [ci, vp, ci, vp, ci, ci, ci, ci, vp, vp, vp]
    lib.slam_ransac_select.restype = ci
    lib.slam_ransac_select.argtypes = [vp, vp, ci, ci, ci, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    lib.slam_refine_poses.restype = ci
    lib.slam_refine_poses.argtypes = [vp] * 6 + [ci, vp, ci, vp, ci, vp, ci, ci, vp, vp, vp]
    lib.slam_refine_winners.restype = ci
    lib.slam_refine_winners.argtypes = [vp, vp, cd] + [vp] * 7 + [ci, vp, ci, vp, ci, ci, vp, vp, vp, vp]
    lib.slam_refine_winners_stereo.restype = ci
    lib.slam_refine_winners_stereo.argtypes = lib.slam_refine_winners.argtypes
    lib.slam_ransac_adopt.restype = ci
    lib.slam_ransac_adopt.argtypes = [ci, vp, ci, ci] + [vp] * 8
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().slam_last_error()
        raise SlamError(f"{what} failed ({status}): {msg.decode() if msg else ''}")


KNN_VARIANTS = {"fp4": 0, "valu": 1}
SCORE_VARIANTS = {"auto": 0, "mixed": 0, "f64": 1, "split": 2, "wide": 3, "mfma": 4}


def select_kernels(knn=None, score=None):
    """Process-wide choice of kernel variants with identical results (slam_select_kernels):
    knn in KNN_VARIANTS, score in SCORE_VARIANTS; None leaves a choice alone."""
    st = load().slam_select_kernels(-1 if knn is None else KNN_VARIANTS[knn], -1 if score is None else SCORE_VARIANTS[score])
    check(st, "slam_select_kernels")
