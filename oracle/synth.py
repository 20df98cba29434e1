"""Re-export of synthdata.opnet (seeded synthetic clips / weights) under its historical name, so that the oracle, the
golden generator and the tests share one generator.  Data only - no model arithmetic lives here."""
from synthdata.opnet import *            # noqa: F401,F403
from synthdata.opnet import T_FRAMES, MAX_OBJECTS, FRAME_SHAPES   # noqa: F401
