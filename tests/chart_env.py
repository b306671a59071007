"""Which Pose3 chart the scenario tests run under: SLIDE_TEST_CHART=expmap switches product AND oracle to SLIDE_CHART_EXPMAP (the
GTSAM_POSE3_EXPMAP behaviour); default: Cayley, what cubeFactor.h:96-97 names.  DESIGN 2: the reference's own evidence does not decide."""
import os


def chart_kw(mod):
    """Keyword arguments for mod.default_params / mod.OrcParams.default (mod = slide_slam_amd or oracle.pyoracle)."""
    return {"pose_chart": mod.CHART_EXPMAP} if os.environ.get("SLIDE_TEST_CHART") == "expmap" else {}


def chart_name():
    return "expmap" if os.environ.get("SLIDE_TEST_CHART") == "expmap" else "cayley"
