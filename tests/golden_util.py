import ast
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def bm_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "bm_*.npz")))


def morph_cases():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "morph_*.npz")))


def load_bm(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    kw = {str(k): ast.literal_eval(str(v)) for k, v in zip(z["param_names"], z["param_values"])}
    return z["left"], z["right"], z["disp"], kw


def load_morph(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
