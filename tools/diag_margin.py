import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tests import codes_margin_worker as W
rng = np.random.default_rng(1)
for params in W.PARAM_SETS[:3] + [W.PARAM_SETS[4]]:
    r = W.one_parameter_set(rng, params, 100000)
    print(json.dumps({k: v for k, v in r.items() if k != 'windows'}))
