"""Entry point with the reference's dispatch (ref: run_me.py:7-32): `python run_me.py {icrl,cpg,gail,run_policy} <flags>`."""
import sys

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in ("icrl", "cpg", "gail", "run_policy"):
        raise SystemExit("usage: python run_me.py {icrl,cpg,gail,run_policy} <flags>   (airl / random_agent / pruning are not part of this build)")
    if sys.argv[1] == "icrl":
        from icrl_amd.icrl import main
    elif sys.argv[1] == "run_policy":
        from icrl_amd.run_policy import main
    elif sys.argv[1] == "cpg":
        from icrl_amd.cpg import main
    else:
        from icrl_amd.gail import main
    main(sys.argv[1:])
