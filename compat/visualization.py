"""`from visualization import create_comprehensive_report, test_visualization_components` (run_multimodal_deer.py:82): plots
are out of scope."""


def create_comprehensive_report(*a, **k):
    raise NotImplementedError("visualization: plotting is outside the mmdeer hot path (SURVEY.md 2)")


def test_visualization_components(*a, **k):
    raise NotImplementedError("visualization: plotting is outside the mmdeer hot path (SURVEY.md 2)")


test_visualization_components.__test__ = False
