"""Model-level entry points around the hot path (SURVEY.md section 8(f) "next" rows): per-architecture layer maps
(the two the BASELINE configs need), `quantize_model`, a packed LUT checkpoint format and the GPTQ-style
perplexity evaluator.  The reference's model zoo / loader / writer (models/base.py, loader.py, writer.py, 52
definitions) are NOT rebuilt: these helpers operate on any Hugging Face style decoder the caller has built."""
from .calibration import as_batches, get_c4, get_wikitext2, load_text_split, wikitext2_test_ids
from .definitions import LAYER_MAPS, layer_map_for
from .quantize import gptq_style_ppl, load_quantized, quantize_model, save_quantized

__all__ = ["LAYER_MAPS", "layer_map_for", "quantize_model", "save_quantized", "load_quantized", "gptq_style_ppl",
           "get_wikitext2", "get_c4", "wikitext2_test_ids", "load_text_split", "as_batches"]
