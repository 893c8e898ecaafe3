"""MI355X-native tabular Q-learning engine behind dist_classicrl's ``OptimalQLearningBase`` /
``BaseRuntime`` API.  The compute path is ``csrc/libqlearn_engine.so`` (hand-written HIP for gfx950);
this package is the thin host-side mirror of the reference interface.  See DESIGN.md."""

__version__ = "0.1.0"
