cd $GRAFT_REPO_ROOT
for nt in 0 1; do
for shape in "64 4096 3072" "64 4096 4096" "64 10112 1024"; do
  echo "== NTLD=$nt stamps $shape"
  ICZ_DEV_NTLD=$nt timeout -k 10 120 python3 tools/perf_skinny_stamps.py $shape 2>&1 | grep -v amdgpu.ids | tail -3
done
done
python3 - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from simpleimagecaptionzoo_amd.butd import gemm
torch.manual_seed(0)
for (M,N,K) in [(64,4096,3072),(64,4096,4096),(64,10112,1024),(40,4096,3072),(64,2048,512+1024)]:
    X=torch.randn(M,K,device='cuda'); W=torch.randn(N,K,device='cuda')*0.03
    C=gemm('nt',X,W,None,0)
    R=(X.double()@W.double().t())
    print(M,N,K,'max rel err', float((C.double()-R).abs().max()/R.abs().max()))
PY
