"""Public predict API: `get_predictions`, `MolNexTRSingleton` (mirrors reference MolNexTR/molnextr.py:42-309).

Same names, arguments and result keys. Differences, on purpose:
  * the model runs on an MI355X through libmolnextr_hip.so only — there is no MPS / CPU detection and NO CPU
    fallback on error (reference molnextr.py:179-189,256-268): errors propagate;
  * the checkpoint is taken from $MOLNEXTR_CHECKPOINT (reference `.pth` or our `.safetensors`); the reference downloads
    molnextr_best.pth at run time (molnextr.py:129-143), which is impossible offline. Without the variable the call
    raises — MOLNEXTR_CHECKPOINT=synthetic opts into the deterministic hash-generated weights (tests only);
  * `bond_sets` IS returned with atoms_bonds=True (the reference computes it but drops it).
"""
import logging
import os

import torch

logger = logging.getLogger("molnextr")
logger.setLevel(os.environ.get("MOLNEXTR_DEBUG", "INFO").upper() if os.environ.get("MOLNEXTR_DEBUG") else logging.INFO)


class MolNexTRSingleton:
    _instance = None
    _device = None
    _device_name = None

    @classmethod
    def get_instance(cls):
        """Get or create the singleton model instance."""
        if cls._instance is None:
            cls._detect_hardware()
            from .model import molnextr
            path = os.environ.get("MOLNEXTR_CHECKPOINT")
            if not path:
                raise RuntimeError("set MOLNEXTR_CHECKPOINT to a MolNexTR checkpoint (reference .pth or .safetensors); "
                                   "MOLNEXTR_CHECKPOINT=synthetic selects the deterministic test weights")
            logger.info("Initializing MolNexTR (%s) on %s", path, cls._device_name)
            cls._instance = molnextr(path, cls._device)
        return cls._instance

    @classmethod
    def get_device(cls):
        if cls._device is None:
            cls._detect_hardware()
        return cls._device, cls._device_name

    @classmethod
    def _detect_hardware(cls):
        if not torch.cuda.is_available():
            raise RuntimeError("molnextr_amd needs an AMD Instinct MI355X (no CPU path in this engine)")
        cls._device = torch.device("cuda", torch.cuda.current_device())
        cls._device_name = f"ROCm GPU: {torch.cuda.get_device_name(cls._device)}"


def get_predictions(imagepath: str, atoms_bonds: bool = False, smiles: bool = True, predicted_molfile: bool = False):
    """Predictions for one chemical-structure image (reference molnextr.py:214-309)."""
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    model = MolNexTRSingleton.get_instance()
    start.record()
    predictions = model.predict_final_results(imagepath, return_atoms_bonds=atoms_bonds)
    result = {}
    if smiles:
        result["predicted_smiles"] = predictions["predicted_smiles"]
    if atoms_bonds:
        result["atom_sets"] = predictions["atom_sets"]
        result["bond_sets"] = predictions["bond_sets"]
    if predicted_molfile:
        result["predicted_molfile"] = predictions["predicted_molfile"]
    result["device_info"] = MolNexTRSingleton.get_device()[1]
    end.record()
    torch.cuda.synchronize()
    result["prediction_time_seconds"] = start.elapsed_time(end) / 1000.0
    if not (smiles or atoms_bonds or predicted_molfile):
        return predictions
    return result
