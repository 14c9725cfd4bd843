import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from glomeruli_segmentation_amd.engine import EspnetEngine
from glomeruli_segmentation_amd.synth import FOLD_MEAN_STD
z = np.load("tests/golden/weights_fold1.npz"); g = np.load("tests/golden/stages_fold1.npz")
eng = EspnetEngine({k: z[k] for k in z.files}, q=0)   # q=0: nothing overwrites the reduced map after the down-sampler
mean, std = FOLD_MEAN_STD[1]
eng.segment(torch.from_numpy(g["tile"][None]).cuda(), mean, std)
torch.cuda.synchronize()
np.save(sys.argv[1], eng.read_stage("level3_reduce"))
