import random, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from types import SimpleNamespace
from oracle import scat_oracle as O
from oracle.util import rel_err
from scat_amd import synth, ops
from scat_amd.models.hand_net import EncoderTransformerHRNet, _TokensFn
from scat_amd import nn as snn
T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
opt = SimpleNamespace(vit_heads=8, vit_depth=3, iteration=3, pos_embed=True, mask_rate=0.2)
net = EncoderTransformerHRNet(opt, T(synth.mean_params(104, 61)))
sdn = synth.hrnet_wrapper_state(105, net.state_dict())
net.load_state_dict(synth.to_torch(sdn), strict=True); net.cuda().train()
sd = synth.to_torch(sdn)
x = T(synth.images(106, 2))
with torch.no_grad():
    f_ref = O.hrnet_forward(sd, x, "main_encoder.", True)
    f = net.main_encoder(x.cuda())
    print("hrnet out", rel_err(f, f_ref), float(f_ref.abs().max()))
    c_ref = F.conv2d(f_ref.reshape(2, 512, 28, 28), sd["conv1x1_channel_reduction.weight"], stride=2, padding=1)
    c = net.conv1x1_channel_reduction(f_ref.cuda().view(2, 512, 28, 28))
    print("conv", rel_err(c, c_ref), float(c_ref.abs().max()))
    masked = list(range(128)); random.seed(7); random.shuffle(masked); masked = masked[:25]
    t_ref = c_ref.reshape(2, 128, -1) + sd["positionalEncoding.pe"][:2]
    t_ref = t_ref.clone(); t_ref[:, masked, :] = sd["mask_token"]
    midx = torch.tensor(masked, dtype=torch.int32, device="cuda")
    t = _TokensFn.apply(c_ref.cuda().view(2, 128, -1), net.positionalEncoding.pe[0], net.mask_token, midx)
    print("tokens", rel_err(t, t_ref), float(t_ref.abs().max()))
    v_ref = O.vit_forward(sd, t_ref, "transformer.", 3, 8)
    v = net.transformer(t_ref.cuda(), None)
    print("vit", rel_err(v, v_ref), float(v_ref.abs().max()))
    m_ref = v_ref.mean(1); m = snn.token_mean(v_ref.cuda())
    print("mean", rel_err(m, m_ref))
    pred = T(synth.mean_params(104, 61)).repeat(2, 1)
    for _ in range(3):
        pred = pred + F.linear(torch.cat([m_ref, pred], -1), sd["regressor.0.weight"], sd["regressor.0.bias"])
    out, _ = ops.regressor_fwd(m_ref.cuda().contiguous(), None, net.mean_params.reshape(-1), net.regressor[0].weight, net.regressor[0].bias, 3, root_relative=False)
    print("head", rel_err(out, pred), float(pred.abs().max()))
    # sensitivity: vit on perturbed tokens
    v2 = O.vit_forward(sd, t_ref * (1 + 1e-6), "transformer.", 3, 8)
    print("vit sensitivity to 1e-6 input scale:", rel_err(v2, v_ref))
