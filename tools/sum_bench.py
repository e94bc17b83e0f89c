import json,sys
for f in sys.argv[1:]:
    try: d=json.load(open(f))
    except Exception as e: print(f, "ERR", e); continue
    s=d.get("secondary",{})
    def g(k):
        v=s.get(k,{})
        x=v.get("ms_per_step") or v.get("ms") or v.get("ms_per_batch")
        return round(x,3) if x else None
    ph={k:round(v,3) for k,v in (d.get("phases_ms") or {}).items()}
    print(f.split('/')[-1], round(d["value"]), round(d["ms_per_step"],3), ph.get('rollouts'), ph.get('backward'), {k:g(k) for k in ("xe_step","xe_step_spatial49","beam5","beam5_b128","aoa_scst_step","scst_step_b8","scst_step_end_biased")})
