"""Round-5 diagnostic: tests #128-#131 of tests/test_trainer_gpu.py in suite order, many times in one process, to catch an order-dependent flake."""
import os, sys, traceback, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_trainer_gpu as T


class MP:
    def __init__(self): self.saved = {}
    def setenv(self, k, v): self.saved.setdefault(k, os.environ.get(k)); os.environ[k] = v
    def delenv(self, k, raising=True): self.saved.setdefault(k, os.environ.get(k)); os.environ.pop(k, None)
    def undo(self):
        for k, v in self.saved.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


bad = 0
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for it in range(n):
    for name in ("test_first_real_update_of_the_transient_head_carries_the_global_bias_correction",
                 "test_update_is_skipped_while_the_status_word_or_the_reduced_fault_flag_is_set",
                 "test_without_radiometric_normalization_matches_the_oracle",
                 "test_lean_step_consumes_the_message_and_matches_the_kept_message_path",
                 "test_export_render_after_native_updates_uses_the_current_weights"):
        fn = getattr(T, name)
        mp = MP()
        try:
            fn(mp) if fn.__code__.co_argcount else fn()
        except Exception:
            bad += 1
            print(f"iteration {it}: {name} FAILED", flush=True)
            traceback.print_exc()
        finally:
            mp.undo()
print("failures:", bad, "in", n, "rounds")
