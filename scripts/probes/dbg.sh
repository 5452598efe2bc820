for d in 0; do echo "DBG=$d"; EMAP_DBG=$d python - <<'PY' 2>&1 | tail -5
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch, emap_amd
from conftest import net_state
kw, state = net_state("d8w256L10")
net = emap_amd.UDFNetwork(scale=1.0, precision="bf16", **kw); net.load_state_dict(state); net = net.cuda()
x = torch.rand(64,3).cuda()
with torch.no_grad(): u,g = net.hip_udf(x, with_grad=True)
torch.cuda.synchronize(); print("fine", float(u.abs().max()))
PY
done
