import sys, json
for line in sys.stdin:
    line=line.strip()
    if line.startswith("{"):
        d=json.loads(line); print({k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k.startswith("attn") or k.startswith("kept")})
    elif line.startswith(("flags","slack","==")): print(line)
