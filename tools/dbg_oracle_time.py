import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import usable_cores
from oracle import rnnt_oracle as ro
from oracle.rnnt_oracle import OracleJointNet, make_batch, training_loss
ro.PER_UTTERANCE = True
print("usable", usable_cores(), "torch default", torch.get_num_threads(), flush=True)
for nt in (int(a) for a in sys.argv[1:]):
    torch.set_num_threads(nt)
    tn = dict(input_size=80, hidden_size=640, output_size=640, num_layers=6, rnn_type="lstm", dropout=0.0, bidirectional=True)
    pn = dict(embedding_size=2048, hidden_size=640, output_size=640, num_layers=1, rnn_type="lstm", dropout=0.0)
    torch.manual_seed(0)
    o = OracleJointNet(dict(tn), dict(pn, pad_token_id=0), 2048).double()
    b = make_batch(2, 1500, 80, 2048, ragged=True, seed=7)
    t0 = time.time(); l = training_loss(o, (b[0].double(),) + b[1:], separable=True); t1 = time.time(); l.backward()
    print("c5 threads", nt, "fwd", round(t1 - t0, 1), "bwd", round(time.time() - t1, 1), flush=True)
