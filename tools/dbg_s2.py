import os, sys, subprocess
sys.path.insert(0, '/root/repo')
if len(sys.argv) > 1:
    import torch
    from preworld_amd import ops, _lib
    D, H, W, cin, cout, c1 = [int(v) for v in sys.argv[1:7]]
    torch.manual_seed(0)
    xf = torch.randn(1, D, H, W, cin, device='cuda')
    x = ops.f32_to_h2(xf)
    w1 = torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05
    ws = [w1] + ([torch.randn(cout, cin, 3, 3, 3, device='cuda') * 0.05] if c1 else [])
    wpk, inv = ops.pack_conv_weights_h2_concat(ws)
    out = ops.conv3d_h2(x, wpk, inv, cout0=cout, cout1=cout if c1 else 0, relu0=True, ksize=3, stride=2)
    torch.cuda.synchronize()
    y0 = out[0] if isinstance(out, (tuple, list)) else out
    y0 = ops.h2_to_f32(y0) if hasattr(ops, 'h2_to_f32') else y0
    ref = torch.relu(torch.nn.functional.conv3d(xf.permute(0, 4, 1, 2, 3), w1, stride=2, padding=1)).permute(0, 2, 3, 4, 1)
    print(sys.argv[1:], _lib.lib().pw_last_kernel().decode(), 'max err', float((y0 - ref).abs().max()), flush=True)
else:
    for cfg in ('4 6 6 32 32 0', '4 6 6 32 32 1', '8 20 20 32 64 1', '8 100 100 64 128 1', '16 40 40 32 64 1', '5 7 9 64 32 1'):
        for mt in ('1', '2'):
            env = dict(os.environ, PW_GATHER_MT=mt)
            r = subprocess.run([sys.executable, __file__] + cfg.split(), env=env, capture_output=True, text=True, timeout=120)
            print('MT', mt, r.stdout.strip() or ('FAIL rc=%d ' % r.returncode + r.stderr.strip()[-300:]), flush=True)
