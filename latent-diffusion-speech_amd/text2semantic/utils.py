"""`get_language_model` with the reference's name and argument convention (reference text2semantic/utils.py:20-28)."""


def get_language_model(**args):
    model_type = args["text2semantic"]["model"]["type"]
    if model_type == "roformer":
        from text2semantic.roformer.roformer import get_model
    else:
        raise ValueError(f" [x] Unknown Model: {model_type}")
    return get_model(args["common"]["n_spk"], **args["text2semantic"])
