"""Drop-in alias: `from models import models` resolves to the MI355X implementation, like the reference's
package of the same name (train_model.py:4, eval_nerf.py)."""
