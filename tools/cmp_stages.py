import json, sys
a=json.load(open(sys.argv[1])); b=json.load(open(sys.argv[2]))
print(f"{sys.argv[1]}: {a['value']} fps {a['ms_per_step']} ms | {sys.argv[2]}: {b['value']} fps {b['ms_per_step']} ms")
for x,y in zip(a['roofline']['stages'], b['roofline']['stages']):
    print(f"  {x['kernel']:42s} {x['ms']:.3f} {x['tflops']:7.1f} | {y['kernel']:38s} {y['ms']:.3f} {y['tflops']:7.1f}")
