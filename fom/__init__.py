"""Drop-in import path of the reference (`from fom.forward_solve import Fin`): thin re-exports of
bayesianinferencedl_amd.fom (repo root on sys.path)."""
