"""MultiKE_CSL.py surface of the reference (code/MultiKE_CSL.py): `MultiKE_CV`, the ITC driver (in-training
combination: every epoch ends with common-space learning).  run_ITC.py's model."""
from __future__ import annotations

from .MultiKE_Late import _ScheduledMultiKE, test, valid  # noqa: F401  (re-exported: run_ITC.py imports them from here)


class MultiKE_CV(_ScheduledMultiKE):
    def __init__(self, data, args, predicate_align_model):
        super().__init__(data, args, predicate_align_model)
        self.flag1, self.flag2, self.early_stop = -1, -1, False
        self.defer_predicate_update = True       # the soft predicate-alignment refresh's host work under the next epoch's kernels
        if hasattr(self.predicate_align_model, "_refresh_device"):
            self.predicate_align_model.device = self.device    # ... and its per-triple work in HBM (no list upload)
        self._define_variables()
        self._define_name_view_graph()
        self._define_relation_view_graph()
        self._define_attribute_view_graph()
        self._define_cross_kg_entity_reference_relation_view_graph()
        self._define_cross_kg_entity_reference_attribute_view_graph()
        self._define_cross_kg_attribute_reference_graph()
        self._define_cross_kg_relation_reference_graph()
        self._define_common_space_learning_graph()

    def run(self):
        """code/MultiKE_CSL.py:36-107."""
        a = self.args
        self._prepare()
        self._test('nv')
        for i in range(1, a.max_epoch + 1):
            print('epoch {}:'.format(i))
            self._train_views(i)
            self.train_common_space_learning_1epo(i, self._entity_list)
            if i >= a.start_valid and i % a.eval_freq == 0:
                self._valid('rv')
                self._valid('av')
                self._valid('final')
                if self.early_stop or i == a.max_epoch:
                    break
            if i >= a.start_predicate_soft_alignment and i % 10 == 0:
                self._update_predicate_alignment()
            self._refresh_neighbours(i)
        self._finish_predicate_update()
        self._save_async = True
        self.save()
        results = {k: self._test(k) for k in ('nv', 'rv', 'av', 'final')}
        self._join_save()
        return results
