#!/usr/bin/env python3
"""The generated Falcon 9 campaign program alone (32,768 rollouts, 1000 ticks per launch, 3 launches) for rocprofv3 --pmc passes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
from elodin_amd.models import falcon9 as f9

fx = f9.AscentExec(f9.sample_params(32768), dtype=np.float32, fast_math=True)
fx.hip.invoke_batch(3000)
fx.close()
print("done")
